"""-m gpu: the snow-MPM step on the HIP path, through the C ABI of include/mpm_hip.h, against the CPU
restatement oracle/mpm_oracle.cpp on the same seeded inputs.

Bars: unknown numbering and num_active bit-exact; the float32 node mass within the accumulation-order noise of a float sum
(the oracle adds in particle order in float32 like the serial reference, the kernel sums in fp64 and rounds once: <= 2e-6
relative at 400 particles per voxel, <= 2e-7 at 8); every fp64 field and particle array <= 1e-4 relative L2 (north_star's
float tolerance) — the measured margins are orders of magnitude inside and asserted much tighter below.
"""
import ctypes

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mo():
    from oracle import mpm_oracle
    return mpm_oracle


def make_pair(fs, mo, pos, vel=None, **kw):
    sim = fs.MpmSim(**kw)
    kept = sim.upload_particles(pos, vel)
    orc = mo.MpmOracle()
    if "transpose_system" in kw:
        orc.set_transposed(kw["transpose_system"])
    assert orc.set_particles(pos, vel) == kept
    return sim, orc


def scene(fs, ppv=400.0, keep=None, seed=0):
    pos = fs.snow_cone(points_per_voxel=ppv, seed=seed)
    if keep:
        pos = pos[np.random.default_rng(7).choice(len(pos), keep, replace=False)]
    return pos


def compare_step(fs, sim, orc, so, tol_mass, tol=1e-9):
    F, P = fs.MPM_F, fs.MPM_P
    assert np.array_equal(sim.field(F.INDICES), orc.field(3))
    assert rel_l2(sim.field(F.CONTAINER), orc.field(0)) < tol_mass
    assert rel_l2(sim.field(F.OUTPUT), orc.field(2)) < tol_mass
    for fid in (F.VEL_BEFORE, F.FORCES, F.VEL):
        e = rel_l2(sim.field(fid), orc.field(fid))
        assert e < max(tol, 20 * tol_mass), (fid, e)
    for what in (P.POS, P.VEL, P.FE, P.FP, P.GRADV, P.VOLUME):
        e = rel_l2(sim.particles(what), orc.particles(what))
        assert e < max(tol, 20 * tol_mass), (what, e)


def test_reference_scene_step_by_step(fs, mo):
    """The reference's own scene (6205 particles, 31^3 grid), 4 steps free-running, every array after every step."""
    sim, orc = make_pair(fs, mo, scene(fs))
    assert sim.num_particles == orc.num_particles == 6205
    # where oracle/_ref travelled, the oracle's solves go through the reference's own solver object (Eigen CG + IncompleteCholesky,
    # mpm.cc:1283): the HIP path is then compared with what that object returns for the assembled matrix
    with_eigen = orc.use_reference_solver()
    print("oracle solves by the reference's Eigen object:", with_eigen)
    for i in range(4):
        sg, so = sim.step(), orc.step()
        assert sg["num_active"] == so["num_active"] > 0
        assert sg["any_active"] == so["any_active"] == 1
        assert sg["cg_error"] < 2.3e-16 and so["cg_error"] < 2.3e-16
        for k in ("dt_in", "dt_out", "max_speed", "max_grad", "max_fp", "max_fe", "max_mi", "max_force_coeff2"):
            assert sg[k] == pytest.approx(so[k], rel=1e-5, abs=1e-300), (i, k)
        assert np.allclose(sg["max_force"], so["max_force"], rtol=1e-5, atol=1e-12)
        compare_step(fs, sim, orc, so, tol_mass=3e-6)
    sim.close()


def test_solution_solves_the_programs_system(fs, mo):
    """x solves A^T x = b for the matrix the restatement assembles the reference's way (std::map of 3x3 blocks): the HIP
    path never forms A, its operator is probed column by column between the two halves of a step."""
    pos = scene(fs, keep=500)
    for transposed in (1, 0):
        sim, orc = make_pair(fs, mo, pos, transpose_system=transposed)
        sim.step(), orc.step()
        sg = sim.step_solve()
        so = orc.step()
        n = 3 * so["num_active"]
        assert sg["num_active"] == so["num_active"]
        rows, cols, vals, b, x = orc.system()
        A = np.zeros((n, n))
        np.add.at(A, (rows, cols), vals)
        Aop = A.T if transposed else A
        bg, xg = sim.system(sg["num_active"])
        # a caller whose buffers were sized for another system is refused before anything is written (mpm_download_system's count)
        guard = np.full(n, 7.0)
        assert fs.lib.mpm_download_system(sim._h, guard.ctypes.data_as(ctypes.c_void_p), guard.ctypes.data_as(ctypes.c_void_p), n - 3) == 1 and (guard == 7.0).all()   # FLUID_ERR_ARG
        with pytest.raises(ValueError):
            sim.system(sg["num_active"] + 1)
        # the node masses carry the float32 accumulation-order noise (~1e-7) into b = v + dt f / m and into A
        assert rel_l2(bg, b) < 2e-6
        assert rel_l2(xg, x) < 2e-6
        assert np.linalg.norm(Aop @ xg - bg) <= 1e-5 * np.linalg.norm(bg)
        # the kernel's own operator: solved to Eigen's tolerance, and equal to the assembled matrix column by column
        assert np.linalg.norm(sim.apply_matrix(xg) - bg) <= 1e-14 * np.linalg.norm(bg)
        cols_g = np.stack([sim.apply_matrix(np.eye(n)[k]) for k in range(n)], axis=1)
        assert np.abs(cols_g - Aop).max() <= 2e-6 * np.abs(Aop).max()
        sim.step_advance()
        compare_step(fs, sim, orc, so, tol_mass=3e-6)
        sim.close()
    # the two systems really differ (rows scaled by 1/m_i)
    assert np.abs(A - A.T).max() > 1e-3


@pytest.mark.parametrize("ppv,keep", [(8.0, None), (40.0, None), (400.0, 900)])
def test_sparser_scenes(fs, mo, ppv, keep):
    """Fewer particles per node: lighter nodes, a stiffer system (more CG iterations), thresholds (mass > 0.1) in play.

    CONVERGENCE, NOT PARITY, wherever the checker's own CG gives up: on these thinned scenes |A - A^T| ~ 1 and the reference's Eigen
    IC-CG stalls at its 2n-iteration cap with a ~1e-8 residual; the restatement's loop can hit the cap too and then takes the DIRECT
    solution of the system (oracle/mpm_oracle.cpp: cg_iters < 0).  A step checked against that states "the kernel's CG converged to
    the exact solution", not "the kernel returns what the reference program would" — printed per step below."""
    sim, orc = make_pair(fs, mo, scene(fs, ppv=ppv, keep=keep, seed=3))
    for i in range(3):
        sg, so = sim.step(), orc.step()
        print(f"ppv {ppv} keep {keep} step {i}: checker {'direct solve (convergence check)' if so['cg_iters'] < 0 else 'CG loop (parity)'}, "
              f"gpu iters {sg['cg_iters']}, checker iters {so['cg_iters']}")
        assert sg["num_active"] == so["num_active"]
        assert sg["cg_error"] < 2.3e-16 and sg["cg_status"] == 1      # converged by Eigen's rule, not stopped at the cap (3) or broken down (2)
        compare_step(fs, sim, orc, so, tol_mass=1e-6, tol=1e-8)
    sim.close()


def test_wall_contact_and_truncated_coordinates(fs, mo):
    """Particles driven into the floor and the side walls: FLIPadvect's ceil/floor rounding, the Coord(int, double, double)
    truncation and the restitution e = 0 (mpm.cc:938-966); particles outside |p| < B - 2 are dropped by add()."""
    rng = np.random.default_rng(11)
    pos = np.concatenate([rng.uniform(-12.9, 12.9, size=(3000, 3)) * [1, 0.02, 1] + [0, -12.6, 0],
                          rng.uniform(11.5, 12.95, size=(500, 3)) * rng.choice([-1, 1], size=(500, 3)),
                          np.array([[13.0, 0, 0], [0, -13.5, 0], [0, 0, 14.9], [12.999, -12.999, 12.999]])])
    vel = rng.normal(size=pos.shape) * 400.0          # up to ~ a cell per step at dt = 1e-3
    sim, orc = make_pair(fs, mo, pos, vel)
    assert sim.num_particles == orc.num_particles == len(pos) - 3
    for i in range(5):
        sg, so = sim.step(), orc.step()
        assert sg["num_active"] == so["num_active"]
        assert sg["dt_out"] == pytest.approx(so["dt_out"], rel=1e-6)
        compare_step(fs, sim, orc, so, tol_mass=1e-6, tol=1e-8)
    v = sim.particles(fs.MPM_P.VEL)
    assert (v == 0).any()                              # some component was zeroed by a wall
    sim.close()


def test_empty_and_call_order(fs):
    sim = fs.MpmSim()
    st = sim.step()                                    # no particles: nothing active, dt stays
    assert st["num_active"] == 0 and st["any_active"] == 0 and st["dt_out"] == 0.001 and st["cg_iters"] == 0
    with pytest.raises(fs.FluidError):
        sim.step_advance()
    sim.step_solve()
    with pytest.raises(fs.FluidError):
        sim.step_solve()
    sim.step_advance()
    with pytest.raises(fs.FluidError):
        sim.apply_matrix(np.zeros(3))
    with pytest.raises(fs.FluidError):
        fs.MpmSim(B=2)
    sim.close()


def test_larger_grid_and_restart(fs, mo):
    """A grid other than the reference's 31^3 (B = 23, W = 21) and a restart from downloaded state (FE, FP, volume, dt)."""
    B, W = 23, 21
    pos = fs.snow_cone(B=B, W=W, layers=8, points_per_voxel=6.0, seed=2)
    sim = fs.MpmSim(B=B, W=W)
    sim.upload_particles(pos)
    orc = mo.MpmOracle(B=B, W=W)
    orc.set_particles(pos)
    for i in range(2):
        sg, so = sim.step(), orc.step()
        assert sg["num_active"] == so["num_active"] > 100
    F, P = fs.MPM_F, fs.MPM_P
    assert np.array_equal(sim.field(F.INDICES), orc.field(3))
    assert rel_l2(sim.particles(P.POS), orc.particles(0)) < 1e-8
    # restart a second handle from the first one's state: the next step is the same step
    sim2 = fs.MpmSim(B=B, W=W)
    sim2.upload_particles(sim.particles(P.POS), sim.particles(P.VEL))
    sim2.set_state(sim.particles(P.FE), sim.particles(P.FP), sim.particles(P.VOLUME), step_no=2)
    sim2.dt = sim.dt
    a, b = sim.step(), sim2.step()
    assert a["num_active"] == b["num_active"]
    for what in (P.POS, P.VEL, P.FE, P.FP):
        assert rel_l2(sim2.particles(what), sim.particles(what)) < 1e-12
    sim.close(), sim2.close()


def test_run_sh_mpm_driver(fs, tmp_path):
    """`./run.sh mpm` (reference contract, run.sh:1-7): builds and runs the driver; the reference's stdout lines per step
    (mpm.cc:1313,1394-1396,417,440-442,1405-1412,584,1418,1427) and its files: simulation/mygrids<i>.vdb with the step's
    output grid, mygrids.vdb with every step's grid (mpm.cc:1383,1434-1437)."""
    import os, subprocess
    from conftest import ROOT
    import vdb_reader
    env = dict(os.environ, MPM_STEPS="3", MPM_OUT=str(tmp_path / "simulation"))
    r = subprocess.run([os.path.join(ROOT, "run.sh"), "mpm"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    per_step = ["DT", "1", "GAH", "DAH", "2", "Max", "GARR", "after", "Error:", "5", "3", "MAX", "4", "DT", "Iteration:"]
    assert len(lines) == 3 * len(per_step) + 1 and lines[-1].startswith("Time Taken")
    for i in range(3):
        blk = lines[len(per_step) * i: len(per_step) * (i + 1)]
        for got, want in zip(blk, per_step):
            assert got.split("\t")[0].split(" ")[0] == want, (i, got, want)
        assert blk[-1] == f"Iteration:\t{i + 1}"
        assert blk[5].startswith("Max Force [") and blk[5].endswith(" 1")
        assert float(blk[8].split()[1]) < 2.3e-16
    assert lines[0] == "DT 0.001"
    # the same run through the binding: the files hold its output grid
    sim = fs.MpmSim()
    sim.upload_particles(fs.snow_cone())
    outs = []
    for i in range(3):
        sim.step()
        outs.append(sim.field(fs.MPM_F.OUTPUT))
    allg = vdb_reader.read(tmp_path / "mygrids.vdb")[1]
    assert len(allg) == 3 and all(g.compression == 3 for g in allg)
    for i in range(3):
        info, grids = vdb_reader.read(tmp_path / "simulation" / f"mygrids{i}.vdb")
        assert len(grids) == 1 and grids[0].type == "Tree_float_5_4_3"
        vals, act = grids[0].dense(-15, 15)
        assert rel_l2(vals, outs[i]) < 3e-6 and outs[i].max() > 0.1
        assert rel_l2(allg[i].dense(-15, 15)[0], outs[i]) < 3e-6
    sim.close()


def test_long_run_through_the_impact(fs):
    """400 steps of the reference's scene: the cone falls at 50 cells per time unit, hits the floor and is compacted.  Every
    solve meets Eigen's stopping rule, the state stays finite and inside the walls, plastic flow shows in det FP."""
    sim = fs.MpmSim()
    sim.upload_particles(fs.snow_cone())
    worst, iters = 0.0, []
    for i in range(400):
        st = sim.step()
        assert st["num_active"] > 0 and np.isfinite(st["max_speed"])
        worst = max(worst, st["cg_error"])
        iters.append(st["cg_iters"])
    assert worst < 2.3e-16 and max(iters) < 200
    P = fs.MPM_P
    pos, FE, FP = sim.particles(P.POS), sim.particles(P.FE), sim.particles(P.FP)
    assert np.isfinite(pos).all() and np.isfinite(FE).all() and np.isfinite(FP).all()
    assert np.abs(pos).max() < 14.0                          # the solid shell starts at |c| > 13
    sv = np.linalg.svd(FE, compute_uv=False)
    assert sv.min() >= 1 - 0.025 - 1e-9 and sv.max() <= 1 + 0.0075 + 1e-9    # the clamp of mpm.cc:548-553 held all along
    assert np.abs(np.linalg.det(FP) - 1).max() > 1e-3       # plastic deformation happened
    sim.close()


def test_device_functions_against_the_references_own_code(fs, mo):
    """mpm_eval: the device functions of the step kernels (polar factors by one-sided Jacobi, stress, the energy Hessian
    applied to a dF, the singular-value clamp) against deformHeader.h's own functions compiled from the reference with its
    vendored Eigen (oracle/_ref/libmpm_ref.so: JacobiSVD, colPivHouseholderQr) — and against the restatement where that
    library did not travel."""
    ref = mo.reference_functions() or mo.restated
    pinned = mo.reference_functions() is not None
    rng = np.random.default_rng(4)

    def rand_F(spread):
        q1, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        q2, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        return q1 @ np.diag(np.exp(rng.uniform(-np.log(spread), np.log(spread), 3))) @ q2.T

    n = 300
    F = np.stack([rand_F(1.03 if k % 2 else 3.0) for k in range(n)] + [np.eye(3), 1.3 * np.eye(3), np.diag([2.0, 2.0, 0.5])])
    R, S = fs.mpm_eval(0, F)
    for k in range(len(F)):
        assert np.abs(R[k] - ref.getR(F[k])).max() < 5e-14 * max(1.0, np.abs(F[k]).max())
        assert np.abs(S[k] - ref.getS(F[k])).max() < 5e-14 * max(1.0, np.abs(F[k]).max())
    mu0, lam0, eps = 48000 / (2 * 1.47), 48000 * 0.47 / (1.47 * 0.06), 10.0
    FE = np.stack([rand_F(1.05) for _ in range(n)])
    FP = np.stack([rand_F(1.05) for _ in range(n)])
    sig, _ = fs.mpm_eval(1, FE, FP, mu0, lam0, eps)
    for k in range(n):
        r = ref.getSigma(mu0, lam0, eps, FE[k], FP[k])
        assert np.abs(sig[k] - r).max() < 1e-11 * np.abs(r).max()
    # the Hessian on the dF of getDelFE: row i = gradW^T F
    lam, mu = 3.1e4, 1.7e4
    g = rng.normal(size=(n, 3))
    for i in range(3):
        dF = np.zeros((n, 3, 3))
        dF[:, i, :] = np.einsum("nr,nrc->nc", g, FE)
        Ap, _ = fs.mpm_eval(2, FE, dF, lam, mu)
        for k in range(0, n, 7):
            r = ref.dPsydFdF(g[k], FE[k], lam, mu, i)
            assert np.abs(Ap[k] - r).max() < 1e-12 * np.abs(r).max()
    tFE = np.stack([rand_F(1.08) for _ in range(n)])
    a, b = fs.mpm_eval(3, tFE, FP, 1 - 0.025, 1 + 0.0075)
    for k in range(n):
        ra, rb = ref.clamp(tFE[k], FP[k], 1 - 0.025, 1 + 0.0075)
        assert np.abs(a[k] - ra).max() < 1e-13 and np.abs(b[k] - rb).max() < 1e-13
    print("device functions pinned by the reference's own code:", pinned)


def test_device_functions_against_committed_reference_outputs(fs):
    """The same device functions against tests/golden/mpm_functions.npz — outputs of the reference's own deformHeader.h code
    (make_mpm_golden.py), committed, so this check does not depend on oracle/_ref having travelled."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpm_functions.npz"))
    mu0, lam0, eps, lam, mu, minv, maxv = z["params"]
    R, S = fs.mpm_eval(0, z["F"])
    scale = np.maximum(1.0, np.abs(z["F"]).max(axis=(1, 2)))[:, None, None]
    assert (np.abs(R - z["R"]) / scale).max() < 5e-14 and (np.abs(S - z["S"]) / scale).max() < 5e-14
    sig, _ = fs.mpm_eval(1, z["FE"], z["FP"], mu0, lam0, eps)
    assert (np.abs(sig - z["sigma"]).max(axis=(1, 2)) / np.abs(z["sigma"]).max(axis=(1, 2))).max() < 1e-11
    for i in range(3):
        dF = np.zeros_like(z["FE"])
        dF[:, i, :] = np.einsum("nr,nrc->nc", z["gradW"], z["FE"])
        Ap, _ = fs.mpm_eval(2, z["FE"], dF, lam, mu)
        ref = z["hessian"][:, i]
        assert (np.abs(Ap - ref).max(axis=(1, 2)) / np.abs(ref).max(axis=(1, 2))).max() < 1e-12
    a, b = fs.mpm_eval(3, z["tFE"], z["FP"], minv, maxv)
    assert np.abs(a - z["clampFE"]).max() < 1e-13 and np.abs(b - z["clampFP"]).max() < 1e-13


def test_solution_matches_the_references_solver_output(fs, mo):
    """tests/golden/mpm_solve_ref_scene.npz: what the reference's own Eigen object returned for step 1 of the reference's scene.
    The HIP path's solution of that step agrees with it to the float32 noise of the node masses."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpm_solve_ref_scene.npz"))
    sim = fs.MpmSim()
    sim.upload_particles(fs.snow_cone())
    sim.step()
    st = sim.step_solve()
    assert st["num_active"] == int(z["num_active"])
    b, x = sim.system(st["num_active"])
    assert rel_l2(b, z["b"]) < 2e-6 and rel_l2(x, z["x_eigen"]) < 2e-6
    sim.step_advance()
    sim.close()


def test_large_upload_round_trip_and_scaled_scene(fs):
    """A large particle set on a fresh handle: what was uploaded comes back bit for bit (the device fills of freshly allocated
    arrays must have finished before the handle's own stream writes them — a race once zeroed 219 k positions now and then),
    and the scaled cone of bench.py's `mpm_scaled` leg keeps thousands of active nodes."""
    B = 63
    pos = fs.snow_cone(B=B, W=B - 2, layers=24, points_per_voxel=64.0, seed=0)
    vel = np.random.default_rng(0).normal(size=pos.shape)
    for _ in range(3):                                   # fresh handles, fresh allocations
        sim = fs.MpmSim(B=B, W=B - 2)
        assert sim.upload_particles(pos, vel) == len(pos)
        assert np.array_equal(sim.particles(fs.MPM_P.POS), pos) and np.array_equal(sim.particles(fs.MPM_P.VEL), vel)
        FE = sim.particles(fs.MPM_P.FE)
        assert np.array_equal(FE, np.broadcast_to(np.eye(3), FE.shape))
        sim.close()
    sim = fs.MpmSim(B=B, W=B - 2)
    sim.upload_particles(pos)
    for _ in range(5):
        st = sim.step()
    assert st["num_active"] > 2500 and st["cg_iters"] > 10 and st["cg_error"] < 2.3e-16
    p = sim.particles(fs.MPM_P.POS)
    assert np.abs(p - pos).max() < 1.0                   # 5 steps at |v| = 50 and dt = 1e-3: a quarter of a cell
    sim.close()


def test_two_runs_give_the_same_bits(fs):
    """The particle -> node sums (transfer, forces, operator) are gathers over cell lists sorted by cell and upload index: no
    atomics, so two runs of the same input agree bit for bit — every array, every step, and the iteration counts (the first
    version summed with fp64 atomics and differed in the last bits from run to run).  The upload order must not matter either
    for the node fields... it does for nothing but the order of the sums, which follows the upload index: same order, same bits."""
    F, P = fs.MPM_F, fs.MPM_P
    pos = fs.snow_cone(B=31, W=29, layers=10, points_per_voxel=48.0, seed=3)
    runs = []
    for _ in range(2):
        sim = fs.MpmSim(B=31, W=29)
        sim.upload_particles(pos)
        st = [sim.step() for _ in range(6)]
        runs.append(([s["cg_iters"] for s in st], [s["num_active"] for s in st],
                     [sim.field(f) for f in (F.CONTAINER, F.VEL, F.FORCES, F.VEL_BEFORE)],
                     [sim.particles(w) for w in (P.POS, P.VEL, P.FE, P.FP, P.GRADV, P.VOLUME)], sim.system()))
        sim.close()
    a, b = runs
    assert a[0] == b[0] and a[1] == b[1] and max(a[0]) > 3
    for x, y in zip(a[2] + a[3], b[2] + b[3]):
        assert np.array_equal(x, y)
    assert all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))
