#!/usr/bin/env python3
"""How scattered is the fluid in the splash?  python tools/splash_topology.py [n] [steps]
Connected components of the fluid cells (6-connectivity) and the bounding box of the largest ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import ndimage
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 440
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
for i in range(steps):
    s = sim.step()
fl = sim.field(fs.FIELD.FLAGS)
fluid = (fl & 2) != 0
print("step", steps, "box", s["box_lo"], s["box_hi"], "unknowns", s["num_active"], "fluid cells", int(fluid.sum()))
lab, nc = ndimage.label(fluid)
sizes = np.bincount(lab.ravel())[1:]
order = np.argsort(-sizes)
print("components", nc, "largest", sizes[order[:5]], "cells in components of size 1:", int((sizes == 1).sum()),
      " <=4:", int(sizes[sizes <= 4].sum()), " <=32:", int(sizes[sizes <= 32].sum()))
big = lab == (order[0] + 1)
idx = np.argwhere(big)
print("largest component bbox", idx.min(0), idx.max(0), "cells", int(big.sum()))
for thr in (4, 32, 256):
    keep = np.isin(lab, np.nonzero(sizes > thr)[0] + 1)
    idx = np.argwhere(keep)
    print(f"components > {thr} cells: bbox", idx.min(0), idx.max(0), "cells", int(keep.sum()), "box cells", int(np.prod(idx.max(0) - idx.min(0) + 1)))
# tiles (8 x 8 x 16 in x, y, z) that hold a fluid cell, with and without the small components
def tiles(mask):
    m = np.zeros(((n + 7) // 8 * 8, (n + 7) // 8 * 8, (n + 15) // 16 * 16), bool)
    m[:n, :n, :n] = mask
    return int(m.reshape(m.shape[0] // 8, 8, m.shape[1] // 8, 8, m.shape[2] // 16, 16).any(axis=(1, 3, 5)).sum())
print("active tiles, all fluid cells:", tiles(fluid))
for thr in (1, 2, 4, 8, 32, 256):
    keep = np.isin(lab, np.nonzero(sizes > thr)[0] + 1)
    print(f"  without components <= {thr} cells: tiles {tiles(keep)}  cells dropped {int(fluid.sum() - keep.sum())}  components dropped {int((sizes <= thr).sum())}")
