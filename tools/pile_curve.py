#!/usr/bin/env python3
"""Largest particle count of a cell along the bench run: python tools/pile_curve.py [n] [steps] [every]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 450
every = int(sys.argv[3]) if len(sys.argv) > 3 else 50
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
lo, hi = fs.grid_bounds(n)
for i in range(1, steps + 1):
    sim.step()
    if i % every == 0:
        p, v = sim.download_particles()
        b = (np.floor(np.abs(p) + 0.5) * np.sign(p) - lo).astype(np.int64)
        u, c = np.unique((b[:, 0] * n + b[:, 1]) * n + b[:, 2], return_counts=True)
        print(f"step {i:4d}: max {c.max():6d}  cells>48 {(c > 48).sum():7d}  cells>128 {(c > 128).sum():6d}  cells>256 {(c > 256).sum():6d}", flush=True)
