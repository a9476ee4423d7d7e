"""-m gpu: the x-slab decomposed step (2 and 3 ranks sharing the one GPU of the box, gloo + host
staging as transport) against the single-GPU step on the same input."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_l2, ROOT

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("world,n,ppc,steps,extra", [(2, 32, 4, 4, ()), (3, 40, 4, 3, ()), (2, 32, 4, 6, ("--uniform", "--vel", "3.0")),
                                                    (2, 32, 4, 4, ("--vel", "1.0", "--blend", "0.9")),
                                                    (2, 32, 4, 3, ("--pile", "600"))])
def test_dist_matches_single(fs, tmp_path, world, n, ppc, steps, extra):
    d = run_dist(world, n, ppc, steps, tmp_path, extra)
    pos = fs.water_cube_drop(n, ppc, seed=0)
    if "--pile" in extra:  # one cell past the P2G form switch: every rank must take the tile form, like the single GPU does
        k = int(extra[extra.index("--pile") + 1])
        pos = np.concatenate([pos, np.round(pos[0]) + np.random.default_rng(5).uniform(-0.4, 0.4, size=(k, 3))])
    vel = None
    if "--vel" in extra:
        vel = np.random.default_rng(1).standard_normal(pos.shape) * float(extra[extra.index("--vel") + 1])
    blend = float(extra[extra.index("--blend") + 1]) if "--blend" in extra else 1.0
    sim = fs.FluidSim(n=n, flip_blend=blend)
    sim.upload_particles(pos, vel)
    st = [sim.step() for _ in range(steps)]
    p, v = sim.download_particles()
    F = fs.FIELD
    # integer work: bit-exact, including the global unknown numbering across ranks
    assert list(d["num_active"]) == [s["num_active"] for s in st]
    assert list(d["outer"]) == [s["outer_passes"] for s in st]
    assert np.array_equal(d["indices"], sim.field(F.INDICES))
    assert len(d["ids"]) == len(pos) and np.array_equal(d["ids"], np.arange(len(pos)))
    # P2G sums have the same order on both paths: container and the pre-solve fields agree to rounding of the solve
    assert rel_l2(d["container"], sim.field(F.CONTAINER)) < 1e-12
    ep, ev = rel_l2(d["pos"], p), rel_l2(d["vel"], v)
    epr = rel_l2(d["pressure"], sim.field(F.PRESSURE))
    print(f"world={world} n={n}: bounds={list(d['bounds'])} counts={list(d['counts'])} pos {ep:.2e} vel {ev:.2e} pressure {epr:.2e} "
          f"iters {list(d['iters'])} vs {[s['cg_iters'] for s in st]} comm calls {list(d['calls'])}")
    assert ep < 1e-9 and ev < 1e-7 and epr < 1e-8
    assert rel_l2(d["velgrid"], sim.field(F.VEL)) < 1e-8
    assert np.allclose(d["dt"], [s["dt_out"] for s in st], rtol=1e-9)


@pytest.mark.parametrize("mode", ["rccl", "device"])
def test_one_rank_over_rccl(fs, tmp_path, mode):
    """World size 1 over the RCCL transports (native ncclAllReduce / torch nccl): all a 1-GPU box can run of them.
    Exercises dlopen + ncclCommInitRank + stream-ordered all-reduces; the neighbour exchange has no peer here."""
    n, ppc, steps = 32, 4, 3
    d = run_dist(1, n, ppc, steps, tmp_path, mode=mode)
    sim = fs.FluidSim(n=n)
    sim.upload_particles(fs.water_cube_drop(n, ppc, seed=0))
    st = [sim.step() for _ in range(steps)]
    p, v = sim.download_particles()
    assert list(d["num_active"]) == [s["num_active"] for s in st]
    assert np.array_equal(d["indices"], sim.field(fs.FIELD.INDICES))
    assert rel_l2(d["pos"], p) < 1e-9 and rel_l2(d["vel"], v) < 1e-7
