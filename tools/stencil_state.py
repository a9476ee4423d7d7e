#!/usr/bin/env python3
"""Does the HBM-proof stencil figure depend on what the process did before?  python tools/stencil_state.py
(1) a fresh process; (2) after a 256^3 simulation has run 30 steps and still holds its memory; (3) after that handle is closed."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
F = fs.FIELD


def measure(tag):
    for prec, T in (("fp64", 8), ("fp32", 4)):
        sim = fs.FluidSim(n=256, precision=prec)
        solid = sim.field(F.SOLID)
        sim.upload_field(F.CONTAINER, (solid == 0).astype(np.float32))
        sim.flags_index()
        sim.upload_field(F.SEARCH, np.random.default_rng(1).uniform(-1, 1, size=(256, 256, 256)) * (solid == 0))
        algo = 256 ** 3 * (2 * T + 1)
        sim.stencil_apply(reps=5, box=0)
        cr = sim.stencil_apply(reps=50, box=0)
        ms = [sim.stencil_apply_hbm(reps=56, box=0, footprint_bytes=1 << 30)[0] for _ in range(3)]
        print(tag, prec, "cache-resident %.3f" % (algo / cr / 1e6 / 8000), "hbm", " ".join("%.3f" % (algo / m / 1e6 / 8000) for m in ms), flush=True)
        sim.close()


measure("fresh   ")
big = fs.FluidSim(n=256)
big.upload_particles(fs.water_cube_drop(256, 8, seed=0))
for _ in range(30):
    big.step()
measure("sim live")
big.close()
measure("closed  ")
