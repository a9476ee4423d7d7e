#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run in the build container).

What they are
  kat_conjgradient_5x5.json   the 5x5 SPD system, right-hand side and expected solution that the
                              reference's own unit test holds (openvdb/unittest/TestConjGradient.cc:58-105),
                              restated as data (numbers only).
  step_n16.npz                inputs (particles) and outputs of ONE step of the oracle on a 16^3 grid
                              with the pressure systems solved by the REFERENCE's solver — the
                              vendored Eigen ConjugateGradient<IncompleteCholesky> compiled from
                              /root/reference/Eigen (oracle/_ref/libeigen_ref.so).
  trace_n24.npz               40-step scalar trace (numActive, passes, dt, error, maxSpeed, sum of
                              container) of the same pipeline on a 24^3 grid + final particle state.
  eigen_icpcg_n20.npz         one assembled pressure system (triplets, b) of a 20^3 scene and the
                              solution/iteration count returned by the vendored Eigen IC-PCG.
The reference program itself (fluid.cc) cannot be built here (needs libopenvdb/TBB/Boost/Half), so
only the solver leg of these fixtures comes from reference code; the rest is the restatement.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

fs = entry.load_package()       # only for the host-side scene generator
oracle = entry.load_oracle()
assert oracle.ref_lib() is not None, "build oracle/_ref first: make -C oracle ref"

# --- KAT restated from the reference's unit test ------------------------------------------------
kat = {
    "source": "openvdb/unittest/TestConjGradient.cc:58-105 (testJacobi)",
    "n": 5,
    "triplets": [[0, 0, 24.0], [0, 2, 6.0], [1, 1, 8.0], [1, 2, 2.0], [2, 0, 6.0], [2, 1, 2.0], [2, 2, 8.0], [2, 3, -6.0],
                 [2, 4, 2.0], [3, 2, -6.0], [3, 3, 24.0], [4, 2, 2.0], [4, 4, 8.0]],
    "b": [1.0, 1.0, 1.0, 1.0, 1.0],
    "expected": [0.0104167, 0.09375, 0.125, 0.0729167, 0.09375],
    "tolerance": 1.0e-5,
    "max_iterations": 20,
}
json.dump(kat, open(os.path.join(HERE, "kat_conjgradient_5x5.json"), "w"), indent=1)

# --- one full step, every field ------------------------------------------------------------------
n = 16
pos = fs.water_cube_drop(n, 3, seed=11)
rng = np.random.default_rng(3)
vel = rng.standard_normal(pos.shape) * 0.7
lo = -(n // 2)
pos[:, 1] += (lo + 2) - pos[:, 1].min() + 0.4   # sit on the floor: wall terms of setRHS are exercised
o = oracle.Oracle(n=n, use_ref_solver=True)
o.set_particles(pos, vel)
st = o.step()
p1, v1 = o.particles()
np.savez_compressed(os.path.join(HERE, "step_n16.npz"), n=n, pos0=pos, vel0=vel, pos1=p1, vel1=v1,
                    container=o.field(0), vel_grid=o.field(2), vel_before=o.field(3), indices=o.field(4), rhs=o.field(5),
                    diver=o.field(6), pressure=o.field(7), num_active=st["num_active"], outer_passes=st["outer_passes"],
                    dt_out=st["dt_out"], error=st["error"], max_speed=st["max_speed"])

# --- scalar trace ----------------------------------------------------------------------------------
n = 24
pos = fs.water_cube_drop(n, 4, seed=0)
o = oracle.Oracle(n=n, use_ref_solver=True)
o.set_particles(pos)
rows = []
for i in range(40):
    s = o.step()
    rows.append([s["num_active"], s["outer_passes"], s["dt_out"], s["error"], s["max_speed"], float(o.field(0).sum(dtype=np.float64))])
pf, vf = o.particles()
np.savez_compressed(os.path.join(HERE, "trace_n24.npz"), n=n, ppc=4, seed=0, trace=np.array(rows), pos_final=pf, vel_final=vf)

# --- one pressure system through the reference's solver -----------------------------------------------
n = 20
pos = fs.water_cube_drop(n, 4, seed=5)
vel = np.random.default_rng(9).standard_normal(pos.shape)
o = oracle.Oracle(n=n)
o.set_particles(pos, vel)
o.p2g(); o.flags_index(); o.rhs_div(); o.build_matrix()
rows_, cols_, vals_, b, _, _ = o.system()
x, it, err = oracle.eigen_icpcg(len(b), rows_, cols_, vals_, b)
np.savez_compressed(os.path.join(HERE, "eigen_icpcg_n20.npz"), n=n, pos=pos, vel=vel, rows=rows_, cols=cols_, vals=vals_, b=b, x=x,
                    iters=it, err=err, eigen_version=oracle.ref_lib().eigen_ref_version().decode())
for f in sorted(os.listdir(HERE)):
    print(f, os.path.getsize(os.path.join(HERE, f)))
