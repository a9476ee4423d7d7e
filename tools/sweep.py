#!/usr/bin/env python3
"""Developer sweep of the marching-stencil knobs: python tools/sweep.py [n] [hbm]
hbm: time the HBM-proof form (launches rotating over > 1 GiB of separate operand sets) instead of back-to-back launches over one set."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hbm = len(sys.argv) > 2 and sys.argv[2] == 'hbm'
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
    for var in (20001, 20002) if len(sys.argv) > 3 and sys.argv[3] == 'probe' else (40402, 40403, 40404, 40602, 40604, 40802, 40803, 40804, 40806, 41402, 41404) if len(sys.argv) > 3 and sys.argv[3] == 'lean' else (60402, 60403, 60404, 60602, 60604, 60802, 60804, 61402, 61404, 70402, 70804) if len(sys.argv) > 3 and sys.argv[3] == 'leany' else (30101, 30102, 30103, 30104, 30201, 30202, 30203, 30401, 30402, 804, 1604, 10402) if len(sys.argv) > 3 else (404, 408, 804, 808, 1604, 1608, 10202, 10204, 10402, 10404, 10408, 10804, 10808, 11604):
        for cx in ((16, 32, 64, 128, 256) if len(sys.argv) > 3 and sys.argv[3] == 'leany' else (8, 16, 32, 64)):
            os.environ["FLUID_MARCH_VARIANT"] = str(var); os.environ["FLUID_MARCH_CX"] = str(cx)
            if hbm:
                ms = min(sim.stencil_apply_hbm(reps=28, box=0, footprint_bytes=1 << 30)[0] for _ in range(2))
            else:
                sim.stencil_apply(reps=3, box=0)
                ms = min(sim.stencil_apply(reps=30, box=0) for _ in range(3))
            print(prec, var, cx, f"{ms*1e3:.1f}us {algo/ms/1e6:.0f} GB/s", flush=True)
    sim.close()
