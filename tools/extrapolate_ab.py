#!/usr/bin/env python3
"""A/B of the extrapolated start of the third and later pressure passes of a step (FLUID_EXTRAPOLATE): python tools/extrapolate_ab.py [n] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 260
for ex in ("1", "0"):
    os.environ["FLUID_EXTRAPOLATE"] = ex
    sim = fs.FluidSim(n=n)
    sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
    its = passes = 0; t0 = time.perf_counter(); per = []
    for i in range(steps):
        st = sim.step(); its += st["cg_iters"]; passes += st["outer_passes"]
        if i == 0: print("  step0", st["cg_iters"], st["outer_passes"])
    p, v = sim.download_particles()
    print(f"extrapolate={ex} n={n} steps={steps} iters={its} passes={passes} time={time.perf_counter()-t0:.2f}s  pos checksum {np.abs(p).sum():.10e}")
    sim.close()
