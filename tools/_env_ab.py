import os, sys, time, json
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as entry
fs = entry.load_package()
n = 256
mark = int(sys.argv[1])
envs = [json.loads(a) for a in sys.argv[2:]]
sim = fs.FluidSim(n=n); sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
for i in range(mark):
    sim.step()
p, v = sim.download_particles()
for env in envs:
    for k, val in env.items(): os.environ[k] = val
    a = fs.FluidSim(n=n); a.upload_particles(p, v)
    a.step(); a.step(); a.step()
    t0 = time.perf_counter(); its = 0; ps = 0
    for k in range(6):
        st = a.step(); its += st["cg_iters"]; ps += st["outer_passes"]
    ms = (time.perf_counter() - t0) / 6 * 1e3
    a.close()
    print(f"state {mark} env {env}: {ms:.2f} ms/step, iters/step {its/6:.1f}, passes/step {ps/6:.2f}, paths {st['paths']}", flush=True)
    for k in env: os.environ.pop(k)
