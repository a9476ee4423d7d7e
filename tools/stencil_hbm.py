#!/usr/bin/env python3
"""The HBM-proof dense stencil sweep alone, for profiling: python tools/stencil_hbm.py <prec> <variant> <cx> [n] [reps]
(variant / cx as FLUID_MARCH_VARIANT / FLUID_MARCH_CX (launch_stencil_march, kernels_stencil.hip): 0 0 = the default)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
prec, var, cx = sys.argv[1], sys.argv[2], sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 28
T = 8 if prec == "fp64" else 4
if var != "0":
    os.environ["FLUID_MARCH_VARIANT"] = var
    os.environ["FLUID_MARCH_CX"] = cx
sim = fs.FluidSim(n=n, precision=prec)
F = fs.FIELD
solid = sim.field(F.SOLID)
sim.upload_field(F.CONTAINER, (solid == 0).astype(np.float32))
sim.flags_index()
sim.upload_field(F.SEARCH, np.random.default_rng(1).uniform(-1, 1, size=(n, n, n)) * (solid == 0))
ms, nsets = sim.stencil_apply_hbm(reps=reps, box=0, footprint_bytes=1 << 30)
algo = n ** 3 * (2 * T + 1)
print(f"{prec} variant {var} cx {cx}: {ms * 1e3:.1f} us per launch, {algo / ms / 1e6:.0f} GB/s algorithmic, {nsets} sets")
sim.close()
