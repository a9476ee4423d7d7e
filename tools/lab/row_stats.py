#!/usr/bin/env python3
"""How many of the row kernel's row visits find an empty row / an empty target column: python tools/lab/row_stats.py [n] step step ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = [int(a) for a in sys.argv[2:]] or [195, 445]
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
done = 0
for st in steps:
    while done < st:
        sim.step(); done += 1
    pos, vel = sim.download_particles()[:2]
    lo, hi = fs.grid_bounds(n)
    c = np.rint(pos).astype(np.int64) - int(lo)       # base cells
    mn, mx = c.min(0) - 1, c.max(0) + 1                 # the P2G box: one cell around the particles
    ext = mx - mn + 1
    zt = 62
    ntz = -(-ext[2] // zt); ztp = -(-ext[2] // ntz)
    occ = np.zeros((ext[0] + 2, ext[1] + 2, ntz), dtype=np.int64)     # particles per (x, y, z piece) with one row of margin
    zp = np.minimum((c[:, 2] - mn[2]) // ztp, ntz - 1)
    np.add.at(occ, (c[:, 0] - mn[0] + 1, c[:, 1] - mn[1] + 1, zp), 1)
    rows = occ[1:-1, 1:-1]
    nb = np.zeros_like(rows)
    for dx in (0, 1, 2):
        for dy in (0, 1, 2):
            nb += occ[dx:dx + ext[0], dy:dy + ext[1]]
    print(f"step {st}: box {ext.tolist()} z pieces {ntz}: row pieces {rows.size}, empty {np.mean(rows == 0):.3f}, with 1-16 particles {np.mean((rows > 0) & (rows <= 16)):.3f}, "
          f"target columns with an empty 3 x 3 neighbourhood {np.mean(nb == 0):.3f}; particles per non-empty row piece: median {np.median(rows[rows > 0]):.0f} mean {rows[rows > 0].mean():.0f}")
