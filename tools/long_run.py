#!/usr/bin/env python3
"""Step time through the whole drop -> splash -> spread sequence: python tools/long_run.py [n] [steps] [particles per cell]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, int(sys.argv[3]) if len(sys.argv) > 3 else 8, seed=0))
t0 = time.perf_counter(); it = 0; passes = 0; tsim = 0.0
for i in range(steps):
    s = sim.step(); it += s["cg_iters"]; passes += s["outer_passes"]; tsim += s["dt_out"]
    if (i + 1) % 50 == 0:
        t1 = time.perf_counter()
        box = [h - l + 1 for l, h in zip(s["box_lo"], s["box_hi"])]
        print(f"steps {i-48:4d}-{i+1:4d}: {(t1-t0)/50*1e3:7.3f} ms/step  t={tsim:6.2f}  box {box}  unknowns {s['num_active']:8d}  iters/solve {it/max(passes,1):5.1f}  passes/step {passes/50:.2f}  dt {s['dt_out']:.4f}", flush=True)
        t0 = t1; it = 0; passes = 0
# per-class timing of the last phase
sim.profile_reset(); sim.profile_enable(1)
for i in range(20):
    s = sim.step()
sim.profile_enable(0)
for name in ("SORT", "P2G", "G2P", "SOLVE", "PCG_SQ", "PCG_XR", "MG_UP0"):
    r = sim.profile_read(getattr(fs.PROF, name))
    if r["sampled"]:
        print(f"  {name:10s} launches/step {r['launches']/20:7.1f}  avg {r['total_ms']/r['sampled']:8.3f} ms  total/step {r['total_ms']/20:8.3f} ms")
import numpy as np
c = sim.field(fs.FIELD.CONTAINER)
p, v = sim.download_particles()
lo, hi = fs.grid_bounds(n)
b = np.floor(np.abs(p) + 0.5) * np.sign(p) - lo
key = (b[:, 0] * n + b[:, 1]) * n + b[:, 2]
u, cnts = np.unique(key.astype(np.int64), return_counts=True)
print("particles per occupied cell: mean", cnts.mean(), "max", cnts.max(), "p99", np.percentile(cnts, 99), "cells>64:", (cnts > 64).sum())
