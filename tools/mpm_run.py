#!/usr/bin/env python3
"""Runs the snow-MPM step for profiling: python tools/mpm_run.py [B] [layers] [points_per_voxel] [steps]
(defaults: the reference's scene, 15 4 400 50).  Prints the per-phase HIP-event times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
fs = entry.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 15
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ppv = float(sys.argv[3]) if len(sys.argv) > 3 else 400.0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
sim = fs.MpmSim(B=B, W=B - 2)
pos = fs.snow_cone(B=B, W=B - 2, layers=layers, points_per_voxel=ppv)
sim.upload_particles(pos)
sts = [sim.step() for _ in range(steps)]
print(f"grid {2 * B + 1}^3 particles {sim.num_particles} steps {steps} unknowns {sts[-1]['num_active']} cg_iters/step {np.mean([s['cg_iters'] for s in sts]):.1f}")
for k in ("ms_transfer", "ms_forces", "ms_solve", "ms_deform", "ms_advect"):
    print(f"  {k[3:]:9s} {np.mean([s[k] for s in sts[2:]]) * 1e3:8.1f} us")
print(f"  apply kernel {np.mean([s['ms_apply_avg'] for s in sts[2:]]) * 1e3:.1f} us")
