#!/usr/bin/env python3
"""Free-running drift of the HIP snow-MPM step against its CPU checker: python tools/mpm_drift.py [points_per_voxel] [steps] [print every]
(test infrastructure: uses oracle/)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
from oracle import mpm_oracle as mo
fs = entry.load_package()
ppv = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
pos = fs.snow_cone(points_per_voxel=ppv)
sim = fs.MpmSim(); sim.upload_particles(pos)
orc = mo.MpmOracle(); orc.set_particles(pos)
for i in range(steps):
    a, b = sim.step(), orc.step()
    if i % every == every - 1 or a["num_active"] != b["num_active"]:
        e = np.linalg.norm(sim.particles(0) - orc.particles(0)) / np.linalg.norm(orc.particles(0))
        print(i, a["num_active"], b["num_active"], f"pos rel {e:.2e}", a["cg_iters"], b["cg_iters"], f"{a['dt_out']:.3e} {b['dt_out']:.3e}", f"{a['max_speed']:.3f} {b['max_speed']:.3f}")
        if a["num_active"] != b["num_active"]: break
