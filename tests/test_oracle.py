"""CPU (-m "not gpu"): pins the oracle — the CPU restatement used as the checker on the GPU box.

 * known-answer vectors restated from the reference's own unit tests (tests/golden/kat_*.json);
 * the committed fixtures made with the reference's solver (vendored Eigen IC-PCG);
 * where oracle/_ref is built (this container), the live vendored Eigen: the restated CG loop must
   take the SAME number of iterations as Eigen's own Jacobi CG and agree with the IC-PCG solution;
 * an analytic hydrostatic column (the fluid-step analogue of TestPoissonSolver.cc:254-309, P = -y).
"""
import json
import os

import numpy as np
import pytest

from conftest import rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_spline_known_values(oracle):
    # fluid.cc:22-37
    assert oracle.spline(0.0) == 1.0
    assert abs(oracle.spline(0.5) - 0.25) < 1e-15
    assert oracle.spline(1.0) == 0.0 and oracle.spline(-1.5) == 0.0
    assert oracle.spline(-0.3) == oracle.spline(0.3)
    # continuous at |x| = 0.5; it is 1.5 x the cubic B-spline of 2x, so its shifts by 1/2 sum to 1.5
    # (NOT a partition of unity on the integer lattice, which is why P2G divides by the weight sum, fluid.cc:1138-1142)
    assert abs(oracle.spline(0.5 - 1e-9) - oracle.spline(0.5 + 1e-9)) < 1e-8
    for x in np.linspace(-0.5, 0.5, 11):
        assert abs(sum(oracle.spline(x - k / 2.0) for k in range(-4, 5)) - 1.5) < 1e-14


def test_kat_conjgradient_5x5(oracle):
    """The reference's own known-answer test for a PCG (TestConjGradient.cc:58-105), as data."""
    k = json.load(open(os.path.join(GOLD, "kat_conjgradient_5x5.json")))
    tr = np.array(k["triplets"])
    rows, cols, vals = tr[:, 0].astype(np.int32), tr[:, 1].astype(np.int32), tr[:, 2]
    x, it, _ = oracle.cg_triplets(k["n"], rows, cols, vals, k["b"])
    assert np.allclose(x, k["expected"], atol=k["tolerance"]) and it <= k["max_iterations"]
    if oracle.ref_lib() is not None:
        for solver in (oracle.eigen_icpcg, oracle.eigen_jacobi_cg):
            xe, ite, _ = solver(k["n"], rows, cols, vals, k["b"])
            assert np.allclose(xe, k["expected"], atol=k["tolerance"]) and ite <= k["max_iterations"]


def test_golden_eigen_system(oracle):
    """A pressure system assembled by the restatement; solution from the reference's IC-PCG (fixture)."""
    g = np.load(os.path.join(GOLD, "eigen_icpcg_n20.npz"))
    o = oracle.Oracle(n=int(g["n"]))
    o.set_particles(g["pos"], g["vel"])
    o.p2g(); o.flags_index(); o.rhs_div(); o.build_matrix()
    rows, cols, vals, b, _, _ = o.system()
    assert np.array_equal(rows, g["rows"]) and np.array_equal(cols, g["cols"])
    assert np.array_equal(vals, g["vals"]) and np.array_equal(b, g["b"])
    o.solve()
    _, _, _, _, _, p = o.system()
    e = rel_l2(p, g["x"])
    print("restated Jacobi-CG vs vendored Eigen IC-PCG (fixture):", e, "iters", o.stats()["cg_iters_last"], "vs", int(g["iters"]))
    assert e < 1e-10


def test_golden_step(oracle):
    """One whole step against the fixture made with the reference's solver in the loop."""
    g = np.load(os.path.join(GOLD, "step_n16.npz"))
    o = oracle.Oracle(n=int(g["n"]))
    o.set_particles(g["pos0"], g["vel0"])
    st = o.step()
    assert st["num_active"] == int(g["num_active"]) and st["outer_passes"] == int(g["outer_passes"])
    assert np.array_equal(o.field(4), g["indices"])           # integer work: exact
    assert np.array_equal(o.field(0), g["container"])         # same serial float32 order: exact
    assert np.array_equal(o.field(3), g["vel_before"])
    assert rel_l2(o.field(7), g["pressure"]) < 1e-9           # Jacobi-CG vs IC-PCG, both to 2.2e-16
    assert rel_l2(o.field(2), g["vel_grid"]) < 1e-10
    p, v = o.particles()
    assert rel_l2(p, g["pos1"]) < 1e-12 and rel_l2(v, g["vel1"]) < 1e-10
    assert abs(st["dt_out"] - float(g["dt_out"])) < 1e-12


def test_golden_trace(oracle):
    g = np.load(os.path.join(GOLD, "trace_n24.npz"))
    tr = g["trace"]
    n = int(g["n"])
    # the scene generator is product code; the trace pins it too (same particles -> same numActive)
    import __graft_entry__ as entry
    fs = entry.load_package()
    pos = fs.water_cube_drop(n, int(g["ppc"]), seed=int(g["seed"]))
    o = oracle.Oracle(n=n)
    o.set_particles(pos)
    for i in range(len(tr)):
        s = o.step()
        assert s["num_active"] == int(tr[i, 0]) and s["outer_passes"] == int(tr[i, 1]), i
        assert abs(s["dt_out"] - tr[i, 2]) <= 1e-9 * tr[i, 2]
        assert abs(s["max_speed"] - tr[i, 4]) <= 1e-8 * max(tr[i, 4], 1e-30)
    p, v = o.particles()
    assert rel_l2(p, g["pos_final"]) < 1e-8 and rel_l2(v, g["vel_final"]) < 1e-6
    # step 0 takes 8 passes, later steps one
    assert int(tr[0, 1]) == 8 and set(tr[3:, 1].astype(int)) == {1}


def test_restated_cg_matches_live_eigen(oracle):
    """Only where oracle/_ref exists: the loop restated from ConjugateGradient.h:28-90 takes exactly
    the iterations Eigen's own DiagonalPreconditioner CG takes, and both agree with the IC-PCG."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref/libeigen_ref.so not built (needs /root/reference)")
    import __graft_entry__ as entry
    fs = entry.load_package()
    for n, ppc in ((16, 3), (24, 4)):
        pos = fs.water_cube_drop(n, ppc, seed=2)
        vel = np.random.default_rng(4).standard_normal(pos.shape)
        o = oracle.Oracle(n=n)
        o.set_particles(pos, vel)
        o.p2g(); o.flags_index(); o.rhs_div(); o.build_matrix(); o.solve()
        rows, cols, vals, b, _, p = o.system()
        xj, itj, _ = oracle.eigen_jacobi_cg(len(b), rows, cols, vals, b)
        xi, iti, _ = oracle.eigen_icpcg(len(b), rows, cols, vals, b)
        assert o.stats()["cg_iters_last"] == itj, (o.stats()["cg_iters_last"], itj)
        assert rel_l2(p, xj) < 1e-13 and rel_l2(p, xi) < 1e-10
        assert iti < itj  # IC really is the stronger preconditioner


def test_hydrostatic_column(oracle):
    """Fluid at rest filling the bottom of the tank, open to air above: the pressure that setRHS/setA/solve
    produce is hydrostatic, p = rho*g*dx * (fluid cells at or above), constant in x,z
    (analytic; cf. the P = -y tank of TestPoissonSolver.cc:254-309)."""
    n = 16
    lo = -(n // 2); hi = lo + n - 1
    wlo, whi = lo + 2, hi - 2
    # particles on cell centres of 3 layers y = wlo+1..wlo+3 over the whole x,z extent of W (4 per cell, tiny jitter)
    rng = np.random.default_rng(0)
    c = np.array([(x, y, z) for x in range(wlo, whi + 1) for y in range(wlo + 1, wlo + 4) for z in range(wlo, whi + 1)], dtype=np.float64)
    pos = np.repeat(c, 4, axis=0) + rng.uniform(-0.2, 0.2, size=(4 * len(c), 3))
    o = oracle.Oracle(n=n)
    o.set_particles(pos)
    o.p2g(); o.flags_index(); o.rhs_div(); o.build_matrix(); o.solve()
    idx = o.field(4)
    p = o.field(7)
    fluid_y = sorted(set(np.nonzero(idx >= 0)[1]))           # index space
    assert fluid_y == list(range(2, 7))                      # y = wlo .. wlo+4 (support reaches one cell beyond)
    g, dx, rho = 10.0, 1.0, 1.0
    top = fluid_y[-1]
    for iy in fluid_y:
        layer = p[2:n - 2, iy, 2:n - 2]
        expect = rho * g * dx * (top - iy + 1)
        assert np.allclose(layer, expect, rtol=1e-6), (iy, layer.min(), layer.max(), expect)


def test_extrapolate_and_resample_restatements():
    """The two unused utilities of the reference, restated in oracle/fluid_oracle.cpp: properties that follow from
    fluid.cc:705-802 and 1053-1080.  Extrapolation: cells P2G wrote keep their value, every other cell inside W gets one, the first
    layer is the mean of its defined 26-neighbours, cells outside W stay 0.  Resampling: no base cell keeps more than the cap, the
    first particles of a cell (index order) stay, the parked ones sit at boundary + 40."""
    from oracle import oracle
    n = 20
    o = oracle.Oracle(n=n)
    rng = np.random.default_rng(2)
    pos = rng.uniform(-3, 3, size=(1500, 3))
    vel = rng.standard_normal((1500, 3))
    o.set_particles(pos, vel)
    o.p2g()
    w, v0 = o.field(1).copy(), o.field(2).copy()
    o.extrapolate()
    v1 = o.field(2)
    src = w > 0
    assert np.array_equal(v1[:, src], v0[:, src])
    inner = np.zeros((n, n, n), bool); inner[2:-2, 2:-2, 2:-2] = True
    assert np.all(np.abs(v1[:, inner]).sum(0) > 0) and np.all(v1[:, ~inner] == 0)
    # one first-layer cell by hand: mean of the source neighbours, scanned x, y, z ascending
    ix, iy, iz = [int(a[0]) for a in np.nonzero(inner & ~src & (np.abs(v0).sum(0) == 0) &
                                                 (np.array([[[src[max(i-1,0):i+2, max(j-1,0):j+2, max(k-1,0):k+2].any() for k in range(n)] for j in range(n)] for i in range(n)])))]
    nb = [(i, j, k) for i in range(ix - 1, ix + 2) for j in range(iy - 1, iy + 2) for k in range(iz - 1, iz + 2) if src[i, j, k]]
    acc = np.zeros(3)
    for q in nb:
        acc = v0[:, q[0], q[1], q[2]] + acc
    assert np.array_equal(v1[:, ix, iy, iz], acc / len(nb))
    o.resample(2)
    p, _ = o.particles()
    lo = -(n // 2)
    cells = np.floor(np.abs(pos) + 0.5) * np.sign(pos)
    kept = p[:, 0] < n
    hi = lo + n - 1
    seen = {}
    for i in range(len(pos)):
        c = tuple(cells[i])
        if c[0] >= hi - 10:                  # the reference's `rx < 50` (boundary - 10): cells beyond are not looked at
            assert kept[i]
            continue
        want = seen.get(c, 0) < 2
        assert kept[i] == want
        if want:
            seen[c] = seen.get(c, 0) + 1
    assert (~kept).sum() > 100 and np.all(p[~kept] == hi + 40)
    assert np.array_equal(p[kept], pos[kept])
