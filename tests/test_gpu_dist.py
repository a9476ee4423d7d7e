"""-m gpu: the 3-D block decomposed step against the single-GPU step on the same input.

Blocks run as host threads of this process over the in-process transport (2 x 2 x 2 on the one GPU of the box: the box
allows few processes per card), plus 2 / 3 ranks as separate processes over gloo (torch transport, staged through the host)
and 1 rank over both RCCL transports.  Integer work (unknown numbering across blocks, outer passes) must be bit-exact;
the decomposed solve runs the globally coupled V-cycle, so its iteration counts must stay close to the one-GPU counts."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_l2, ROOT

pytestmark = pytest.mark.gpu


def scene(fs, n, ppc, vel=0.0, pile=0):
    pos = fs.water_cube_drop(n, ppc, seed=0)
    if pile:  # one cell past the P2G form switch: every rank must take the tile form, like the single GPU does
        pos = np.concatenate([pos, np.round(pos[0]) + np.random.default_rng(5).uniform(-0.4, 0.4, size=(pile, 3))])
    v = None
    if vel:
        v = np.random.default_rng(1).standard_normal(pos.shape) * vel
    return pos, v


def run_blocks(fs, dims, n, pos, vel, steps, mode, uniform=False, solid=None, rebalance=None, **kw):
    """dims[0] x dims[1] x dims[2] blocks as threads of this process; returns the assembled result."""
    fd = fs.load_dist()
    size = dims[0] * dims[1] * dims[2]
    cuts = fd.uniform_cuts(n, dims) if uniform else fd.partition_blocks(n, pos, dims)
    grp = fd.LocalGroup(size)
    F = fs.FIELD
    sims = [None] * size

    def work(r):
        sim = fd.DistFluidSim(n, dims, cuts, grp.comms[r], dist_solve=mode, **kw)
        sims[r] = sim
        if solid is not None:
            sim.set_solid(solid)              # the GLOBAL array on every rank; each keeps its window
        sim.upload_global(pos, vel)
        if rebalance:
            sim.set_rebalance(*rebalance)
        st = [sim.step() for _ in range(steps)]
        p, v, ids = sim.download_local()
        return dict(st=st, p=p, v=v, ids=ids, idx=sim.field(F.INDICES), cont=sim.field(F.CONTAINER), pres=sim.field(F.PRESSURE),
                    vel=sim.field(F.VEL), cuts=sim.cuts, moved=sim.n_rebalanced, info=sim.info())

    try:
        res = grp.run(work)
    finally:
        for s in sims:
            if s is not None:
                s.close()
        grp.close()
    ids = np.concatenate([r["ids"] for r in res])
    o = np.argsort(ids)
    out = dict(ids=ids[o], pos=np.concatenate([r["p"] for r in res])[o], vel=np.concatenate([r["v"] for r in res])[o],
               st=res[0]["st"], all_st=[r["st"] for r in res], cuts=res[0]["cuts"], counts=[len(r["ids"]) for r in res],
               moved=[r["moved"] for r in res], all_cuts=[r["cuts"] for r in res], info=[r["info"] for r in res])
    for k, key in (("indices", "idx"), ("container", "cont"), ("pressure", "pres"), ("velgrid", "vel")):
        out[k] = fd.assemble(n, sims, [r[key] for r in res])
    return out


def single(fs, n, pos, vel, steps, solid=None, **kw):
    sim = fs.FluidSim(n=n, **kw)
    if solid is not None:
        sim.set_solid(solid)
    sim.upload_particles(pos, vel)
    st = [sim.step() for _ in range(steps)]
    p, v = sim.download_particles()
    F = fs.FIELD
    out = dict(st=st, pos=p, vel=v, indices=sim.field(F.INDICES), container=sim.field(F.CONTAINER), pressure=sim.field(F.PRESSURE),
               velgrid=sim.field(F.VEL))
    sim.close()
    return out


def compare(d, ref, npart, label, tol_p=1e-9, tol_v=1e-7, tol_pr=1e-8):
    st, rs = d["st"], ref["st"]
    # every rank reports the same global scalars
    for other in d["all_st"]:
        assert [s["num_active"] for s in other] == [s["num_active"] for s in st]
        assert [s["cg_iters"] for s in other] == [s["cg_iters"] for s in st]
        assert [s["outer_passes"] for s in other] == [s["outer_passes"] for s in st]
    # integer work: bit-exact, including the global unknown numbering across blocks
    assert [s["num_active"] for s in st] == [s["num_active"] for s in rs]
    assert [s["outer_passes"] for s in st] == [s["outer_passes"] for s in rs]
    assert np.array_equal(d["indices"], ref["indices"])
    assert len(d["ids"]) == npart and np.array_equal(d["ids"], np.arange(npart))
    # P2G sums have the same order on both paths
    assert rel_l2(d["container"], ref["container"]) < 1e-12
    ep, ev = rel_l2(d["pos"], ref["pos"]), rel_l2(d["vel"], ref["vel"])
    epr = rel_l2(d["pressure"], ref["pressure"])
    print(f"{label}: cuts={d['cuts']} counts={d['counts']} pos {ep:.2e} vel {ev:.2e} pressure {epr:.2e} "
          f"iters {[s['cg_iters'] for s in st]} vs {[s['cg_iters'] for s in rs]} passes {[s['outer_passes'] for s in st]}")
    assert ep < tol_p and ev < tol_v and epr < tol_pr
    assert rel_l2(d["velgrid"], ref["velgrid"]) < 1e-8
    assert np.allclose([s["dt_out"] for s in st], [s["dt_out"] for s in rs], rtol=1e-9)
    return st, rs


CASES = [
    # dims, n, ppc, steps, scene kw, sim kw
    ((2, 1, 1), 32, 4, 4, {}, {}),
    ((1, 1, 2), 32, 4, 4, {}, {}),
    ((3, 1, 1), 40, 4, 3, {}, {}),
    ((2, 2, 1), 40, 4, 3, {"vel": 1.0}, {}),
    ((2, 2, 2), 48, 4, 4, {"vel": 2.0}, {}),
    ((2, 2, 2), 40, 4, 3, {"vel": 1.0}, {"flip_blend": 0.9}),
    ((2, 1, 1), 32, 4, 3, {"pile": 600}, {}),
    ((2, 2, 1), 50, 4, 3, {"vel": 0.5}, {}),          # grid sizes that are no multiple of 4 (the reference's own is 121)
    ((2, 1, 2), 33, 3, 3, {}, {}),
]


@pytest.mark.parametrize("mode", ["decomposed", "replicated"])
@pytest.mark.parametrize("dims,n,ppc,steps,skw,kw", CASES)
def test_blocks_match_single(fs, mode, dims, n, ppc, steps, skw, kw):
    pos, vel = scene(fs, n, ppc, **skw)
    ref = single(fs, n, pos, vel, steps, **kw)      # (both start their solves from the previous pressure by default)
    d = run_blocks(fs, dims, n, pos, vel, steps, mode, **kw)
    st, rs = compare(d, ref, len(pos), f"{mode} {dims} n={n}")
    if mode == "replicated":
        # the same arithmetic on every rank as on one GPU
        assert rel_l2(d["pos"], ref["pos"]) < 1e-14 and [s["cg_iters"] for s in st] == [s["cg_iters"] for s in rs]
    else:
        # globally coupled V-cycle: the iteration count must not grow with the number of blocks (another hierarchy
        # anchoring than the one-GPU run: a few iterations either way)
        a, b = sum(s["cg_iters"] for s in st), sum(s["cg_iters"] for s in rs)
        assert a <= 1.2 * b + 2 * sum(s["outer_passes"] for s in rs), (a, b)


@pytest.mark.parametrize("split,gather", [(1, "exchange"), (2, "exchange"), (1, "allreduce"), (2, "allreduce")])
def test_decomposed_levels_on_blocks(fs, split, gather, monkeypatch):
    """Both forms of the coupled V-cycle: level 1 gathered (split 1) and level 1 on the blocks with halo exchanges (split 2);
    the gathered level assembled by the blocks sending their coarse cells to each other (default up to 2 x 2 x 2) or by a SUM
    all-reduce of the zero-padded level."""
    monkeypatch.setenv("FLUID_DIST_SPLIT", str(split))
    monkeypatch.setenv("FLUID_DIST_GATHER", gather)
    n, ppc, steps = 64, 4, 3
    pos, vel = scene(fs, n, ppc, vel=1.0)
    ref = single(fs, n, pos, vel, steps)
    d = run_blocks(fs, (2, 2, 2), n, pos, vel, steps, "decomposed")
    st, rs = compare(d, ref, len(pos), f"split {split} {gather}")
    a, b = sum(s["cg_iters"] for s in st), sum(s["cg_iters"] for s in rs)
    assert a <= 1.2 * b + 2 * sum(s["outer_passes"] for s in rs), (a, b)


def test_decomposed_from_zero_start(fs):
    """solve_start = zero on both sides: every solve from x0 = 0 like the reference's cg.solve(b) (the iteration counts the reference's)."""
    n, ppc, steps = 40, 4, 3
    pos, vel = scene(fs, n, ppc, vel=1.0)
    ref = single(fs, n, pos, vel, steps, solve_start="zero")
    d = run_blocks(fs, (2, 2, 2), n, pos, vel, steps, "decomposed", solve_start="zero")
    st, rs = compare(d, ref, len(pos), "x0 = 0, 2x2x2")
    a, b = sum(s["cg_iters"] for s in st), sum(s["cg_iters"] for s in rs)
    assert a <= 1.2 * b + 2 * sum(s["outer_passes"] for s in rs), (a, b)


def test_decomposed_jacobi_and_uniform_cuts(fs):
    """Eigen's diagonal preconditioner over the blocks: iteration for iteration the one-GPU Jacobi PCG (the sums differ by rounding only)."""
    n, ppc, steps = 32, 4, 2
    pos, vel = scene(fs, n, ppc)
    ref = single(fs, n, pos, vel, steps, preconditioner="jacobi")
    d = run_blocks(fs, (2, 2, 1), n, pos, vel, steps, "decomposed", uniform=True, preconditioner="jacobi")
    st, rs = compare(d, ref, len(pos), "jacobi 2x2x1")
    assert all(abs(a["cg_iters"] - b["cg_iters"]) <= 2 * b["outer_passes"] for a, b in zip(st, rs))


@pytest.mark.parametrize("mode", ["decomposed", "replicated"])
def test_obstacle_across_the_cuts(fs, mode):
    """A solid block inside W (the reference's commented 'big wall', fluid.cc:1333-1345) that straddles the cut planes: Neumann
    faces inside the domain, in every block's window; the cube lands on it."""
    n, steps = 32, 12
    sim0 = fs.FluidSim(n=n)
    solid = sim0.field(fs.FIELD.SOLID).copy()
    sim0.close()
    solid[4:28, 2:9, 12:20] = 1
    pos = fs.water_cube_drop(n, 4, seed=2); pos[:, 1] -= 4.0
    ref = single(fs, n, pos, None, steps, solid=solid)
    d = run_blocks(fs, (2, 1, 2), n, pos, None, steps, mode, uniform=True, solid=solid)
    compare(d, ref, len(pos), f"obstacle {mode}", tol_p=1e-8, tol_v=1e-6, tol_pr=1e-7)


@pytest.mark.parametrize("slack", [None, "64"])
def test_block_empties_and_fills(fs, slack, monkeypatch):
    """A block that holds no particle at first (uniform cuts, the cube in one corner region) and receives them as the fluid falls.
    With almost no spare capacity (FLUID_DIST_SLACK) the particle arrays and the routing buffers must GROW on demand: a rank
    never fails alone for lack of room (its peers would wait for it in the next exchange)."""
    if slack:
        monkeypatch.setenv("FLUID_DIST_SLACK", slack)
    n, steps = 48, 12
    pos, _ = scene(fs, n, 4)
    pos = pos + np.array([6.0, 9.0, 0.0])        # off-centre: the low-y blocks start empty
    ref = single(fs, n, pos, None, steps)
    d = run_blocks(fs, (1, 2, 2), n, pos, None, steps, "decomposed", uniform=True)
    assert min(d["counts"]) >= 0
    compare(d, ref, len(pos), "fills", tol_p=1e-8, tol_v=1e-6, tol_pr=1e-7)


def test_overlapped_halo_exchange_changes_nothing(fs, monkeypatch):
    """The residual's halo exchange runs on a second stream behind the level-0 down-leg tiles that read no received cell
    (FLUID_DIST_OVERLAP, on by default): same tiles, same arithmetic — bit-identical to the serial order, iteration for
    iteration, and both equal to one GPU."""
    n, steps = 72, 5
    pos, vel = scene(fs, n, 4, vel=0.5)
    ref = single(fs, n, pos, vel, steps)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FLUID_DIST_OVERLAP", flag)
        out[flag] = run_blocks(fs, (2, 2, 2), n, pos, vel, steps, "decomposed")
    a, b = out["1"], out["0"]
    assert [s["cg_iters"] for s in a["st"]] == [s["cg_iters"] for s in b["st"]]
    assert np.array_equal(a["pressure"], b["pressure"]) and np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["vel"], b["vel"])
    compare(a, ref, len(pos), "overlap", tol_p=1e-8, tol_v=1e-6, tol_pr=1e-7)


@pytest.mark.parametrize("mode", ["decomposed", "replicated"])
def test_cut_planes_follow_the_water(fs, mode):
    """Re-balancing: uniform cuts through an off-centre cube leave some blocks with nothing; with fluid_dist_set_rebalance the
    planes are placed anew by particle count (every rank the same planes, every particle to its new owner, a new window per
    rank) and the run goes on to the same answer as one GPU — with an obstacle (the new windows need the global solid array again)
    and a moving state."""
    n, steps = 64, 12
    pos, vel = scene(fs, n, 4, vel=0.3)
    pos = pos + np.array([7.0, 9.0, -5.0])
    solid = np.zeros((n, n, n), dtype=np.uint8)
    solid[:2] = solid[-2:] = 1; solid[:, :2] = solid[:, -2:] = 1; solid[:, :, :2] = solid[:, :, -2:] = 1
    solid[20:30, 2:10, 24:40] = 1                    # a block on the floor
    ref = single(fs, n, pos, vel, steps, solid=solid)
    before = run_blocks(fs, (2, 2, 2), n, pos, vel, 1, mode, uniform=True, solid=solid)
    d = run_blocks(fs, (2, 2, 2), n, pos, vel, steps, mode, uniform=True, solid=solid, rebalance=(4, 1.3))
    print(f"rebalance {mode}: counts {before['counts']} -> {d['counts']}, cuts {before['cuts']} -> {d['cuts']}, moved {d['moved']}")
    assert all(m >= 1 for m in d["moved"]) and len(set(d["moved"])) == 1          # every rank, equally often
    assert all(c == d["all_cuts"][0] for c in d["all_cuts"]) and d["cuts"] != before["cuts"]
    assert max(d["counts"]) < max(before["counts"])                                # the fullest block got lighter
    compare(d, ref, len(pos), f"rebalance {mode}", tol_p=1e-8, tol_v=1e-6, tol_pr=1e-7)


@pytest.mark.parametrize("shape", ["sheet", "needle", "blobs", "corner", "odd"])
def test_one_allreduce_per_iteration_gives_the_same_solve(fs, shape, monkeypatch):
    """The decomposed PCG in its Chronopoulos-Gear form (FLUID_DIST_CG=cgear, the default: w = A z, ONE all-reduce of {|r|^2, r.z, w.z}
    per iteration, s and q = A s by recurrence) against the loop of ConjugateGradient.h:28-90 as the one-GPU solve runs it
    (FLUID_DIST_CG=cg: two scalar all-reduces per iteration) and against one GPU, on the awkward domains of
    test_solve_on_awkward_domains cut into 2 x 2 x 2 blocks: Eigen's stopping rule is still met (relres < 2.3e-16), the iteration
    count grows by at most 10 % (+ 1), the pressure is the same."""
    from test_gpu_parity import _shape_particles
    n, steps = 48, 3
    rng = np.random.default_rng(5)
    pos = _shape_particles(fs, n, shape, rng)
    vel = rng.standard_normal(pos.shape) * 0.5
    ref = single(fs, n, pos, vel, steps)
    out = {}
    for form in ("cg", "cgear"):
        monkeypatch.setenv("FLUID_DIST_CG", form)
        out[form] = run_blocks(fs, (2, 2, 2), n, pos, vel, steps, "decomposed", uniform=True)
        compare(out[form], ref, len(pos), f"{shape} {form}", tol_p=1e-8, tol_v=1e-6, tol_pr=1e-7)
        assert all(s["relres"] < 2.3e-16 for s in out[form]["st"]), [s["relres"] for s in out[form]["st"]]
        assert all(i["cg_form"] == (form == "cgear") for i in out[form]["info"])
        assert all(i["overlap"] == 1 for i in out[form]["info"])      # the overlapped halo exchange passed its one-time check against the serial one on every rank
    a, b = sum(s["cg_iters"] for s in out["cg"]["st"]), sum(s["cg_iters"] for s in out["cgear"]["st"])
    print(f"{shape}: iterations cg {a} cgear {b}")
    assert b <= 1.1 * a + steps
    assert rel_l2(out["cgear"]["pressure"], out["cg"]["pressure"]) < 1e-9


def test_a_rank_that_cannot_build_its_new_window_keeps_every_rank_on_the_old_planes(fs, monkeypatch):
    """Re-balancing needs a second window per rank for a moment.  When ONE rank cannot build it (FLUID_DIST_FAIL_REBUILD names the
    rank), every rank must take the same way out — drop the attempt, keep the old planes — and the FOLLOWING steps must run on all of
    them: no rank fails alone later on a stale error, nobody waits for a peer.  The answer is the one of a run that never re-balanced."""
    monkeypatch.setenv("FLUID_DIST_FAIL_REBUILD", "3")
    n, steps = 48, 10
    pos, vel = scene(fs, n, 4, vel=0.3)
    pos = pos + np.array([5.0, 7.0, -3.0])
    ref = single(fs, n, pos, vel, steps)
    d = run_blocks(fs, (2, 2, 2), n, pos, vel, steps, "decomposed", uniform=True, rebalance=(3, 1.1))
    assert d["moved"] == [0] * 8                                  # every attempt was given up, on every rank
    assert len({i["rebalances_refused"] for i in d["info"]}) == 1 and d["info"][0]["rebalances_refused"] >= 1
    assert all(c == d["all_cuts"][0] for c in d["all_cuts"])
    monkeypatch.delenv("FLUID_DIST_FAIL_REBUILD")
    plain = run_blocks(fs, (2, 2, 2), n, pos, vel, steps, "decomposed", uniform=True)
    assert d["cuts"] == plain["cuts"]
    assert np.array_equal(d["pos"], plain["pos"]) and np.array_equal(d["pressure"], plain["pressure"])   # the refused attempts changed nothing
    compare(d, ref, len(pos), "refused re-balance", tol_p=1e-8, tol_v=1e-6, tol_pr=1e-7)


def test_a_rank_that_cannot_grow_fails_every_rank(fs, monkeypatch):
    """The one allocation a step can still need is room for particles that migrate in.  When ONE rank cannot get it
    (FLUID_DIST_FAIL_GROW names the rank), every rank must leave that step with an error — the one that failed with its own,
    the others with FLUID_ERR_PEER — instead of waiting for it in the next exchange for ever.  The threads here do NOT wake
    each other (no fluid_local_group_abort): each has to return by itself."""
    import threading
    monkeypatch.setenv("FLUID_DIST_SLACK", "64")
    monkeypatch.setenv("FLUID_DIST_FAIL_GROW", "1")
    fd = fs.load_dist()
    n, steps, dims = 48, 12, (1, 2, 2)
    pos, _ = scene(fs, n, 4)
    pos = pos + np.array([6.0, 9.0, 0.0])        # off-centre: the low-y blocks start empty and must grow when the water arrives
    cuts = fd.uniform_cuts(n, dims)
    grp = fd.LocalGroup(4)
    sims, result = [None] * 4, [None] * 4

    def work(r):
        sim = fd.DistFluidSim(n, dims, cuts, grp.comms[r], dist_solve="decomposed")
        sims[r] = sim
        sim.upload_global(pos)
        for i in range(steps):
            try:
                sim.step()
            except Exception as e:  # noqa: BLE001
                result[r] = (i, getattr(e, "code", None), str(e))
                return
        result[r] = (steps, 0, "")

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=180)
    hung = [r for r, x in enumerate(th) if x.is_alive()]
    if hung:
        fd.lib.fluid_local_group_abort(grp.handle)     # let the stuck threads go before failing
    assert not hung, f"ranks {hung} are still waiting for a peer that has returned"
    print(result)
    assert len({res[0] for res in result}) == 1 and result[0][0] < steps       # all of them, in the same step
    assert result[1][1] == 2 and "refused" in result[1][2]                      # FLUID_ERR_HIP where it happened
    assert all(result[r][1] == 5 and "another rank failed" in result[r][2] for r in (0, 2, 3))   # FLUID_ERR_PEER elsewhere
    for s in sims:
        if s is not None:
            s.close()
    grp.close()


@pytest.mark.parametrize("n,ppc", [(256, 8), (512, 4)])
def test_baseline_multi_gpu_configs_on_blocks(fs, n, ppc):
    """BASELINE.json configs[3] (256^3 domain-decomposed 2 x 2 x 2) and configs[4] (512^3, 4 particles per cell, 8 GPUs) AT SIZE,
    the eight blocks as threads over the in-process transport on this box's one GPU: two whole steps (the 8-pass first step
    and a steady one) against the one-GPU run — numbering bit-exact, fields to rounding, iteration counts within 20 %."""
    pos = fs.water_cube_drop(n, ppc, seed=0)
    steps = 2
    ref = single(fs, n, pos, None, steps)
    d = run_blocks(fs, (2, 2, 2), n, pos, None, steps, "decomposed")
    st, rs = compare(d, ref, len(pos), f"decomposed 2x2x2 n={n}")
    a, b = sum(s["cg_iters"] for s in st), sum(s["cg_iters"] for s in rs)
    assert a <= 1.2 * b, (a, b)


def _pool_with_spray(fs, n, rng):
    """A shallow pool wall to wall, and closed pockets above it: single cells and pairs, some of them lying across the cut planes of a
    2 x 2 x 2 decomposition (those must stay in the global solve), none touching another."""
    lo, hi = fs.grid_bounds(n)
    w0, w1 = lo + 2, hi - 2
    def fill(cells, ppc):
        cells = np.asarray(cells, dtype=np.float64).reshape(-1, 3)
        return np.repeat(cells, ppc, axis=0) + rng.uniform(-0.3, 0.3, size=(len(cells) * ppc, 3))
    pool = np.stack(np.meshgrid(np.arange(w0, w1 + 1), np.arange(w0, w0 + 5), np.arange(w0, w1 + 1), indexing="ij"), -1).reshape(-1, 3)
    parts = [fill(pool, 2)]
    drops = []
    for x in range(w0 + 3, w1 - 3, 6):
        for y in range(w0 + 12, w1 - 3, 7):
            for z in range(w0 + 3, w1 - 3, 6):
                drops.append((x, y, z))
                if (x + y + z) % 3 == 0:
                    drops.append((x + 1, y, z))            # a two-cell pocket; x = -1 gives one across the x cut
    drops += [(-1, 10, 5), (0, 10, 5), (7, -1, -9), (7, 0, -9), (11, 13, -1), (11, 13, 0)]   # pockets across each cut plane (cells -1 | 0)
    parts.append(fill(sorted(set(drops)), 3))
    return np.concatenate(parts)


def test_decomposed_with_droplets_and_galerkin_levels(fs, monkeypatch):
    """Round 3's solver pieces in the decomposed step: closed pockets are found per rank (owned cells only: a pocket across a cut stays in
    the global solve), leave the system on both sides of every cut (count-byte halo from the owners) and are solved by their owner; the
    coarse levels are Galerkin operators by aggregation, level 1's coefficients gathered from the owners; the level-0 legs and the w = A z
    sweep run over each rank's own lists of active tiles (the down leg's interior / boundary lists cut from them).  Forced on at test size
    (FLUID_TILE_LISTS=1, FLUID_MG_GALERKIN=2); against one GPU with the same pieces on and with them off: same unknown numbering, same
    pressure, iteration counts within 10 %."""
    n, steps = 64, 3
    pos = _pool_with_spray(fs, n, np.random.default_rng(11))
    plain = single(fs, n, pos, None, steps)
    monkeypatch.setenv("FLUID_TILE_LISTS", "1")
    monkeypatch.setenv("FLUID_MG_GALERKIN", "2")
    monkeypatch.setenv("FLUID_DROPLETS_MIN", "0")        # (one GPU: search in every step, also while few pockets are found)
    ref = single(fs, n, pos, None, steps)
    assert all(s["paths"] & 64 for s in ref["st"]) and all(s["paths"] & 128 for s in ref["st"]), [s["paths"] for s in ref["st"]]
    d = run_blocks(fs, (2, 2, 2), n, pos, None, steps, "decomposed", uniform=True)
    compare(d, ref, len(pos), "droplets + galerkin", tol_p=1e-8, tol_v=1e-6, tol_pr=1e-7)
    assert rel_l2(d["pressure"], plain["pressure"]) < 1e-7
    assert all(s["paths"] & 128 for s in d["st"])                                   # Galerkin levels on every rank
    assert sum(1 for st in d["all_st"] if all(s["paths"] & 2 for s in st)) >= 4    # active-tile lists on the ranks that hold unknowns
    assert sum(1 for st in d["all_st"] if st[0]["paths"] & 64) >= 4                  # the ranks above the pool found pockets
    assert not any(s["paths"] & 256 for st in d["all_st"] for s in st)               # every pocket's own CG met the stopping rule
    a, b = sum(s["cg_iters"] for s in d["st"]), sum(s["cg_iters"] for s in ref["st"])
    print(f"droplets + galerkin: iterations decomposed {a} one GPU {b} (plain cycle, droplets in: {sum(s['cg_iters'] for s in plain['st'])})")
    assert abs(a - b) <= 0.1 * b + 2 * steps


@pytest.mark.parametrize("lists", [False, True])
def test_decomposed_through_the_splash(fs, lists, monkeypatch):
    """2 x 2 x 2 blocks, 160 free-running steps: the cube falls across the cut planes, hits the floor, splashes into all eight
    blocks (migration in every direction, ghosts at edges and corners, blocks whose share grows several-fold: buffers grow on
    demand).  No particle is lost or duplicated, every solve converges, and the run stays on the one-GPU trajectory for as
    long as two float-identical-to-rounding runs can (the first steps exactly in the integers)."""
    if lists:   # the mostly-air forms pinned on at test size: active-tile lists and (decomposed: per rank) the droplet search
        monkeypatch.setenv("FLUID_TILE_LISTS", "1")
        monkeypatch.setenv("FLUID_DROPLETS_MIN", "0")
    n, ppc, steps = 48, 4, 160
    pos, _ = scene(fs, n, ppc)
    pos = pos + np.array([3.0, 6.0, -2.0])       # off-centre: unequal blocks
    ref = single(fs, n, pos, None, steps)
    d = run_blocks(fs, (2, 2, 2), n, pos, None, steps, "decomposed", uniform=True)
    assert np.array_equal(d["ids"], np.arange(len(pos)))                       # nobody lost, nobody twice
    assert np.isfinite(d["pos"]).all() and np.isfinite(d["vel"]).all()
    lo, hi = fs.grid_bounds(n)
    assert d["pos"].min() > lo and d["pos"].max() < hi
    na, nb = [s["num_active"] for s in d["st"]], [s["num_active"] for s in ref["st"]]
    same = next((i for i in range(steps) if na[i] != nb[i]), steps)
    print(f"splash: unknown counts identical for the first {same} of {steps} steps; last {na[-1]} vs {nb[-1]}; "
          f"iterations {sum(s['cg_iters'] for s in d['st'])} vs {sum(s['cg_iters'] for s in ref['st'])}; counts {d['counts']}")
    assert same >= 40
    assert abs(na[-1] - nb[-1]) <= 0.02 * nb[-1]
    assert all(s["relres"] < 1e-14 for s in d["st"])
    a, b = sum(s["cg_iters"] for s in d["st"]), sum(s["cg_iters"] for s in ref["st"])
    assert a <= 1.25 * b, (a, b)
    assert max(d["counts"]) > 0 and sum(d["counts"]) == len(pos)


def run_dist(world, n, ppc, steps, tmp_path, extra=(), mode="staged"):
    out = str(tmp_path / f"dist_{world}_{n}_{mode}.npz")
    port = 29500 + (os.getpid() % 2000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), "--grid", str(n), "--ppc", str(ppc),
           "--steps", str(steps), "--mode", mode, "--out", out, *extra]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return np.load(out)


@pytest.mark.parametrize("world,solve", [(2, "decomposed"), (3, "decomposed"), (2, "replicated")])
def test_processes_over_gloo(fs, tmp_path, world, solve):
    """One process per block over torch.distributed (gloo, staged through the host): the transport bench.py falls back to."""
    n, ppc, steps = 32, 4, 3
    d = run_dist(world, n, ppc, steps, tmp_path, ("--solve", solve))
    pos, _ = scene(fs, n, ppc)
    ref = single(fs, n, pos, None, steps)
    assert list(d["num_active"]) == [s["num_active"] for s in ref["st"]]
    assert list(d["outer"]) == [s["outer_passes"] for s in ref["st"]]
    assert np.array_equal(d["indices"], ref["indices"])
    assert np.array_equal(d["ids"], np.arange(len(pos)))
    print(f"gloo world={world} {solve}: comm calls {list(d['calls'])} iters {list(d['iters'])} vs {[s['cg_iters'] for s in ref['st']]}")
    assert rel_l2(d["pos"], ref["pos"]) < 1e-9 and rel_l2(d["vel"], ref["vel"]) < 1e-7 and rel_l2(d["pressure"], ref["pressure"]) < 1e-8


@pytest.mark.parametrize("mode", ["rccl", "device"])
def test_one_rank_over_rccl(fs, tmp_path, mode):
    """World size 1 over the RCCL transports (native ncclAllReduce / torch nccl): all a 1-GPU box can run of them.
    Exercises dlopen + ncclCommInitRank + stream-ordered all-reduces; the neighbour exchange has no peer here."""
    n, ppc, steps = 32, 4, 3
    d = run_dist(1, n, ppc, steps, tmp_path, ("--solve", "decomposed"), mode=mode)
    pos, _ = scene(fs, n, ppc)
    ref = single(fs, n, pos, None, steps)
    assert list(d["num_active"]) == [s["num_active"] for s in ref["st"]]
    assert np.array_equal(d["indices"], ref["indices"])
    assert rel_l2(d["pos"], ref["pos"]) < 1e-9 and rel_l2(d["vel"], ref["vel"]) < 1e-7


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("mode", ["rccl", "device"])
@pytest.mark.parametrize("world,solve", [(2, "decomposed"), (2, "replicated"), (4, "decomposed"), (8, "decomposed")])
def test_real_peers_over_rccl(fs, tmp_path, world, solve, mode):
    """One process per GPU with real peers: grouped ncclSend / ncclRecv between blocks and ncclAllReduce over xGMI, through the
    library's own RCCL transport (`rccl`) and through torch's (`device`).  Skipped on a box with fewer GPUs than ranks (the
    builder's boxes have one): it wakes up on the first multi-GPU box and checks the run against one GPU before bench.py does."""
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs, {_gpus()} visible")
    n, ppc, steps = (64, 4, 4) if world <= 2 else (96, 4, 4)
    d = run_dist(world, n, ppc, steps, tmp_path, ("--solve", solve), mode=mode)
    pos, _ = scene(fs, n, ppc)
    ref = single(fs, n, pos, None, steps)
    assert list(d["num_active"]) == [s["num_active"] for s in ref["st"]]
    assert list(d["outer"]) == [s["outer_passes"] for s in ref["st"]]
    assert np.array_equal(d["indices"], ref["indices"])
    assert np.array_equal(d["ids"], np.arange(len(pos)))
    assert rel_l2(d["pos"], ref["pos"]) < 1e-9 and rel_l2(d["vel"], ref["vel"]) < 1e-7 and rel_l2(d["pressure"], ref["pressure"]) < 1e-8
