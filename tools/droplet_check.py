#!/usr/bin/env python3
"""The droplets a step took out of the global solve against scipy's labelling of the same flags:
python tools/droplet_check.py [n] [steps] [every]   — every claimed set must be a whole connected component of <= 64 unknowns,
claimed once."""
import os, sys
import numpy as np
from scipy import ndimage
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
every = int(sys.argv[3]) if len(sys.argv) > 3 else 25
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
bad_total = 0
for i in range(steps):
    s = sim.step()
    if (i + 1) % every or not (s["paths"] & 64):
        continue
    cells = sim.droplets()
    unk = sim.field(fs.FIELD.INDICES).reshape(-1) >= 0
    lab, ncomp = ndimage.label(unk.reshape(n, n, n))
    lab = lab.reshape(-1)
    sizes = np.bincount(lab)
    seen = np.zeros(unk.size, bool)
    bad = 0
    for c in cells:
        c = c[c >= 0]
        l = lab[c]
        ok = len(c) > 0 and unk[c].all() and (l == l[0]).all() and sizes[l[0]] == len(c) and len(c) <= 64 and not seen[c].any()
        seen[c] = True
        if not ok:
            bad += 1
            if bad <= 3:
                print("   bad droplet:", len(c), "cells, labels", np.unique(l), "true sizes", sizes[np.unique(l)], "cells", c[:8])
    small = ((sizes[1:] <= 64)).sum()
    print(f"step {i+1}: droplets {len(cells)} ({(cells >= 0).sum()} cells) of {small} components <= 64 cells; bad {bad}; dt {s['dt_out']:.4f}", flush=True)
    bad_total += bad
print("BAD" if bad_total else "ok")
