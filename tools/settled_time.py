#!/usr/bin/env python3
"""Per-class times in the settled pool: python tools/settled_time.py [n] [steps before] — runs the bench scene that far, then 10 profiled steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 450
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
for _ in range(pre):
    sim.step()
import time
sim.profile_reset(); sim.profile_enable(1)
t0 = time.perf_counter(); it = 0
for _ in range(10):
    it += sim.step()["cg_iters"]
t1 = time.perf_counter()
sim.profile_enable(0)
out = [f"{(t1 - t0) / 10 * 1e3:.2f} ms/step, {it / 10:.1f} iterations/step"]
for name in ("SORT", "P2G", "G2P", "SOLVE", "PCG_SQ", "PCG_XR", "MG_UP0"):
    r = sim.profile_read(getattr(fs.PROF, name))
    if r["sampled"]:
        out.append(f"{name} {r['total_ms'] / r['sampled'] * 1e3:.0f}us")
print("  ".join(out))
