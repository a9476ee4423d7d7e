#!/usr/bin/env python3
"""Developer sweep of the marching-stencil knobs: python tools/sweep.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for prec, T in (("fp64", 8), ("fp32", 4)):
    sim = fs.FluidSim(n=n, precision=prec)
    F = fs.FIELD
    solid = sim.field(F.SOLID)
    sim.upload_field(F.CONTAINER, (solid == 0).astype(np.float32))
    sim.flags_index()
    s = np.random.default_rng(1).uniform(-1, 1, size=(n, n, n)) * (solid == 0)
    sim.upload_field(F.SEARCH, s)
    algo = n ** 3 * (2 * T + 1)
    # < 10000: one cell per lane (k_stencil_march); 10000 + MY*100 + MD: 16 bytes per lane (k_stencil_vec)
    for var in (804, 1604, 10202, 10204, 10402, 10404, 10408, 10804, 10808, 11604):
        for cx in (8, 16, 32, 64):
            os.environ["FLUID_MARCH_VARIANT"] = str(var); os.environ["FLUID_MARCH_CX"] = str(cx)
            sim.stencil_apply(reps=3, box=0)
            ms = min(sim.stencil_apply(reps=30, box=0) for _ in range(3))
            print(prec, var, cx, f"{ms*1e3:.1f}us {algo/ms/1e6:.0f} GB/s", flush=True)
    sim.close()
