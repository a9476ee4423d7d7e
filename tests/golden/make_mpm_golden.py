#!/usr/bin/env python3
"""Generates tests/golden/mpm_functions.npz and mpm_solve_ref_scene.npz (run in the build container, where
oracle/_ref/libmpm_ref.so and libeigen_ref.so can be built from /root/reference).

mpm_functions.npz      inputs and outputs of the REFERENCE's own constitutive functions (deformHeader.h:22-36, 38-88, 107-249,
                       273-313 and mpm.cc:25-41, compiled as they are against the vendored Eigen): splines on a sweep, getR / getS,
                       getSigma, dPsydFdF for i = 0..2: reference output, nothing in it comes from the restatement.  The
                       singular-value clamp (clampFE / clampFP) is NOT the reference's code: mpm.cc:543-555 sits inside
                       updateDeformationGradient, which takes OpenVDB grids and cannot be built here, so oracle/mpm_ref.cpp
                       (mpm_ref_clamp) is a builder-written wrapper of the same lines around the vendored Eigen::JacobiSVD —
                       those two arrays pin Eigen's SVD under that wrapper, not the program.
mpm_solve_ref_scene.npz  the linear system of step 1 of the reference's scene as the restatement assembles it (triplets, b) and
                       the solution returned by the REFERENCE's solver object (ConjugateGradient<SparseMatrix<double>,
                       Lower|Upper, IncompleteCholesky<double>>, mpm.cc:1283) — the evidence that the program solves A^T x = b.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
from oracle import mpm_oracle as mo  # noqa: E402

fs = entry.load_package()       # only for the host-side scene generator
REF = mo.reference_functions()
assert REF is not None, "build oracle/_ref first: make -C oracle ref"


def rand_F(rng, spread):
    q1, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    q2, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return q1 @ np.diag(np.exp(rng.uniform(-np.log(spread), np.log(spread), 3))) @ q2.T


rng = np.random.default_rng(2024)
xs = np.concatenate([np.linspace(-2.5, 2.5, 1001), [0.5, -0.5, 1.0, -1.0, 1.5, 2.0 ** -40]])
n = 64
F = np.stack([rand_F(rng, 1.03 if k % 2 else 3.0) for k in range(n)])
FE = np.stack([rand_F(rng, 1.05) for _ in range(n)])
FP = np.stack([rand_F(rng, 1.05) for _ in range(n)])
tFE = np.stack([rand_F(rng, 1.08) for _ in range(n)])
g = rng.normal(size=(n, 3))
mu0, lam0, eps = 48000 / (2 * 1.47), 48000 * 0.47 / (1.47 * 0.06), 10.0
lam, mu = 3.1e4, 1.7e4
minv, maxv = 1 - 0.025, 1 + 0.0075
cl = [REF.clamp(tFE[k], FP[k], minv, maxv) for k in range(n)]
np.savez_compressed(
    os.path.join(HERE, "mpm_functions.npz"),
    xs=xs, spline=np.array([REF.spline(x) for x in xs]), spline2=np.array([REF.spline2(x) for x in xs]),
    spline_gradient=np.array([REF.spline_gradient(x) for x in xs]),
    F=F, R=np.stack([REF.getR(f) for f in F]), S=np.stack([REF.getS(f) for f in F]),
    FE=FE, FP=FP, params=np.array([mu0, lam0, eps, lam, mu, minv, maxv]),
    sigma=np.stack([REF.getSigma(mu0, lam0, eps, FE[k], FP[k]) for k in range(n)]),
    gradW=g, hessian=np.stack([[REF.dPsydFdF(g[k], FE[k], lam, mu, i) for i in range(3)] for k in range(n)]),
    tFE=tFE, clampFE=np.stack([c[0] for c in cl]), clampFP=np.stack([c[1] for c in cl]))

orc = mo.MpmOracle()
orc.set_particles(fs.snow_cone())
assert orc.use_reference_solver()
orc.step()
st = orc.step()
rows, cols, vals, b, x = orc.system()
np.savez_compressed(os.path.join(HERE, "mpm_solve_ref_scene.npz"), rows=rows, cols=cols, vals=vals, b=b, x_eigen=x,
                    cg_iters=st["cg_iters"], cg_error=st["cg_error"], num_active=st["num_active"])
print("written:", [f for f in os.listdir(HERE) if f.startswith("mpm_")])
