#!/usr/bin/env python3
"""Droplets on / off from the SAME late states of the drop scene (the run itself is chaotic: two runs that differ in the last
bit part ways within a hundred steps, so whole-run times compare different splashes):
python tools/droplet_ab.py [n] step [step ...]   — pressure difference after one step, ms/step over the next few."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1])
marks = [int(a) for a in sys.argv[2:]]
os.environ["FLUID_DROPLETS"] = "0"
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
for i in range(max(marks) + 1):
    if i in marks:
        p, v = sim.download_particles()
        res = {}
        for mode in ("1", "0"):
            os.environ["FLUID_DROPLETS"] = mode
            a = fs.FluidSim(n=n); a.upload_particles(p, v)
            a.step(); st = a.step()                     # (the tile lists, and the droplets with them, start with the second step)
            pr = a.field(fs.FIELD.PRESSURE).reshape(-1).copy()
            nd = len(a.droplets()) if mode == "1" else 0
            a.step()
            t0 = time.perf_counter()
            its = 0
            for k in range(5):
                its += a.step()["cg_iters"]
            ms = (time.perf_counter() - t0) / 5 * 1e3
            res[mode] = (st, pr, nd, ms, its / 5)
            a.close()
        os.environ["FLUID_DROPLETS"] = "0"
        (sa, pa, nd, msa, ia), (sb, pb, _, msb, ib) = res["1"], res["0"]
        print(f"state {i}: droplets {nd}  |dp| max {np.abs(pa - pb).max():.2e} of {np.abs(pb).max():.1f}  unknowns {sb['num_active']}  "
              f"ms/step on {msa:.2f} off {msb:.2f} ({(1 - msa / msb) * 100:+.1f} %)  iters/step {ia:.0f} {ib:.0f}", flush=True)
    sim.step()
