#!/usr/bin/env python3
"""How much of the late pressure system is spray: python tools/spray_stats.py [n] [steps]
Connected components (6-neighbourhood) of the unknowns after `steps` steps of the drop scene, and how many solver tiles
(4 x 8 x 32 cells) hold nothing but small components."""
import os, sys
import numpy as np
from scipy import ndimage
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 450
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
for i in range(steps):
    s = sim.step()
idx = sim.field(fs.FIELD.INDICES).reshape(n, n, n)
unk = idx >= 0
lab, ncomp = ndimage.label(unk)
sizes = np.bincount(lab.ravel())[1:]
print(f"step {steps}: unknowns {unk.sum()}  components {ncomp}  largest {sizes.max()}  box {s['box_lo']}..{s['box_hi']}  cg_iters {s['cg_iters']}")
for k in (1, 2, 4, 8, 16, 64, 256, 4096):
    print(f"  components of <= {k:5d} cells: {(sizes <= k).sum():7d} components, {sizes[sizes <= k].sum():8d} unknowns")
csize = np.zeros(ncomp + 1, np.int64); csize[1:] = sizes
cell_comp = csize[lab]                      # size of the component each unknown belongs to (0 = not an unknown)
def tiles(a, red):                          # reduce over 4 x 8 x 32 tiles
    t = a.reshape(n // 4, 4, n // 8, 8, n // 32, 32)
    return red(red(red(t, 5), 3), 1)
any_unk = tiles(unk, np.max)
biggest = tiles(cell_comp, np.max)
print(f"  tiles with an unknown: {any_unk.sum()} of {any_unk.size}")
for k in (1, 8, 64, 256, 4096):
    print(f"    of them holding only components of <= {k:5d} cells: {(any_unk & (biggest <= k)).sum()}")
main = lab == (1 + np.argmax(sizes))
print(f"  tiles touched by the largest component: {tiles(main, np.max).sum()}")
# what a windowed search would catch: components of <= 64 cells whose bounding box is at most E cells long on every axis
objs = ndimage.find_objects(lab)
ext = np.array([[sl.stop - sl.start for sl in o] for o in objs])
rows_all = unk.reshape(n, n, n // 32, 32).max(3).sum()
for E in (3, 5, 9, 13):
    ok = (sizes <= 64) & (ext.max(1) <= E)
    keep = np.ones(ncomp + 1, bool); keep[1:] = ~ok; keep[0] = False
    rest = keep[lab]
    print(f"  extent <= {E:2d}: {ok.sum():6d} components, {sizes[ok].sum():7d} unknowns; without them: tiles {tiles(rest, np.max).sum()} (of {any_unk.sum()}), "
          f"z rows of 32: {rest.reshape(n, n, n // 32, 32).max(3).sum()} (of {rows_all}), 8x8x16 tiles {rest.reshape(n//8,8,n//8,8,n//16,16).max(5).max(3).max(1).sum()} (of {unk.reshape(n//8,8,n//8,8,n//16,16).max(5).max(3).max(1).sum()})")
# tile shapes (x, y, z) and the cells they sweep for the unknowns of this state (y is up)
def swept(a, tx, ty, tz):
    t = a.reshape(n // tx, tx, n // ty, ty, n // tz, tz).max(5).max(3).max(1)
    return int(t.sum()) * tx * ty * tz
for shape in ((4, 8, 32), (8, 4, 32), (4, 4, 32), (8, 8, 16), (8, 4, 16), (16, 4, 16), (8, 2, 32), (16, 2, 32), (4, 8, 64), (8, 4, 64)):
    print(f"  tiles {shape}: cells swept {swept(unk, *shape):9d} for {int(unk.sum())} unknowns; without the small components {swept(rest, *shape):9d}")
# how much lower would the solver box be without the droplets?  (y is up; box of all unknowns against the box of what stays in the global solve)
ys = np.nonzero(unk.any(axis=(0, 2)))[0]
for E in (5,):
    ok = (sizes <= 64) & (ext.max(1) <= E)
    keep = np.ones(ncomp + 1, bool); keep[1:] = ~ok; keep[0] = False
    rest = keep[lab]
    yr = np.nonzero(rest.any(axis=(0, 2)))[0]
    ym = np.nonzero(main.any(axis=(0, 2)))[0]
    cnt_by_y = rest.sum(axis=(0, 2))
    print(f"  y range of all unknowns {ys.min()}..{ys.max()}, without the droplets {yr.min()}..{yr.max()}, of the largest component {ym.min()}..{ym.max()}; "
          f"unknowns above y = 40 that stay: {int(cnt_by_y[41:].sum())}, above 60: {int(cnt_by_y[61:].sum())}")
