#!/usr/bin/env python3
"""Particle-per-cell statistics of the bench scene after some steps: python tools/pile_stats.py [n] [steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 450
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
for _ in range(steps):
    sim.step()
p, v = sim.download_particles()
lo, hi = fs.grid_bounds(n)
b = (np.floor(np.abs(p) + 0.5) * np.sign(p) - lo).astype(np.int64)
key = (b[:, 0] * n + b[:, 1]) * n + b[:, 2]
u, c = np.unique(key, return_counts=True)
edges = [1, 9, 17, 33, 49, 65, 129, 257, 1025, 10**9]
for a, e in zip(edges[:-1], edges[1:]):
    m = (c >= a) & (c < e)
    print(f"cells with {a:5d}..{min(e - 1, 99999):5d} particles: {m.sum():8d} cells, {c[m].sum():9d} particles")
x = u // (n * n); y = (u // n) % n; z = u % n
px = np.bincount(x, weights=c, minlength=n)
print("per x-plane: mean", px[px > 0].mean(), "max", px.max(), "at", px.argmax(), " top5", np.sort(px)[-5:])
row = np.bincount(x * n + y, weights=c, minlength=n * n)
print("per (x,y) row: mean", row[row > 0].mean(), "max", row.max(), "rows > 4096:", (row > 4096).sum(), "rows > 1024:", (row > 1024).sum())
py = np.bincount(y, weights=c, minlength=n)
print("per y-plane top5", np.sort(py)[-5:], "argmax", py.argmax(), "ymin", y.min())
big = c > 256
print("monster cells by x:", np.bincount(x[big], minlength=n).nonzero()[0][:20], "by y:", np.unique(y[big])[:20], "by z:", np.unique(z[big])[:20])
