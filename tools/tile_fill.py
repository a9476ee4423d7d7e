#!/usr/bin/env python3
"""How much of the active box the unknowns occupy at several tile granularities: python tools/tile_fill.py [n] [steps]
(fraction of tiles that hold an unknown, and the unknowns' share of the cells of those tiles)."""
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
    st = sim.step()
idx = sim.field(fs.FIELD.INDICES)
lo, hi = st["box_lo"], st["box_hi"]
u = idx[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] >= 0
print(f"box {u.shape}, unknowns {u.sum()} of {u.size} cells ({100 * u.mean():.1f} %)")
for t in ((8, 8, 16), (4, 8, 32), (4, 8, 16), (4, 4, 16), (4, 4, 8), (2, 2, 32), (1, 1, 32), (1, 1, 16), (1, 1, 8)):
    s = [(d + k - 1) // k * k for d, k in zip(u.shape, t)]
    p = np.zeros(s, dtype=bool); p[:u.shape[0], :u.shape[1], :u.shape[2]] = u
    c = p.reshape(s[0] // t[0], t[0], s[1] // t[1], t[1], s[2] // t[2], t[2]).sum(axis=(1, 3, 5))
    act = c > 0
    print(f"tile {t}: {100 * act.mean():5.1f} % of the tiles active, {100 * c[act].sum() / (act.sum() * np.prod(t)):5.1f} % of their cells unknowns")
