"""Worker of the multi-process tests: one block of a decomposed run, one process per block.

  python -m torch.distributed.run --nproc-per-node P --master-addr 127.0.0.1 --master-port PORT \
      tests/dist_worker.py --grid 32 --ppc 4 --steps 3 --mode staged --out /tmp/x.npz
mode staged = gloo + host staging, every rank on GPU 0 (what a 1-GPU box can run);
mode device = torch nccl (RCCL) callbacks, mode rccl = RCCL inside the library; one GPU per rank.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", dest="n", type=int, default=32)
    ap.add_argument("--ppc", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--mode", default="staged")
    ap.add_argument("--solve", default="decomposed")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.mode in ("device", "rccl"):
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        dev = local
    else:
        dist.init_process_group("gloo")
        dev = 0
    fs = entry.load_package()
    fd = fs.load_dist()
    pos = fs.water_cube_drop(a.n, a.ppc, seed=0)
    dims = fd.default_dims(world)
    cuts = fd.partition_blocks(a.n, pos, dims)
    if a.mode == "rccl":
        comm = fd.RcclComm()
    else:
        comm = fd.TorchComm(mode=a.mode, device=torch.device("cuda", dev))
    sim = fd.DistFluidSim(a.n, dims, cuts, comm, device=dev, dist_solve=a.solve)
    sim.upload_global(pos)
    stats = [sim.step() for _ in range(a.steps)]
    p, v, ids = sim.download_local()
    F = fs.FIELD
    blk = {k: sim.field(f) for k, f in (("idx", F.INDICES), ("pres", F.PRESSURE))}
    gathered = [None] * world
    dist.gather_object((p, v, ids, blk, (sim.own_lo, sim.own_hi), comm.calls), gathered if rank == 0 else None, dst=0)
    if rank == 0:
        P = np.concatenate([g[0] for g in gathered]); V = np.concatenate([g[1] for g in gathered]); I = np.concatenate([g[2] for g in gathered])
        o = np.argsort(I)
        n = a.n
        idx = np.zeros((n, n, n), dtype=np.int32); pres = np.zeros((n, n, n))
        for g in gathered:
            lo, hi = g[4]
            sl = tuple(slice(lo[k], hi[k]) for k in range(3))
            idx[sl] = g[3]["idx"]; pres[sl] = g[3]["pres"]
        np.savez(a.out, pos=P[o], vel=V[o], ids=I[o], indices=idx, pressure=pres,
                 num_active=np.array([s["num_active"] for s in stats]), outer=np.array([s["outer_passes"] for s in stats]),
                 iters=np.array([s["cg_iters"] for s in stats]), dt=np.array([s["dt_out"] for s in stats]),
                 calls=np.array([gathered[0][5]["exchange"], gathered[0][5]["allreduce"]]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
