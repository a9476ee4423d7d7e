"""Worker of the multi-process tests: one rank of an x-slab decomposed run.

  python -m torch.distributed.run --nproc-per-node P --master-addr 127.0.0.1 --master-port PORT \
      tests/dist_worker.py --grid 32 --ppc 4 --steps 3 --mode staged --out /tmp/x.npz
mode staged = gloo + host staging, every rank on GPU 0 (what a 1-GPU box can run);
mode device = nccl (RCCL), one GPU per rank.
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
    ap.add_argument("--out", required=True)
    ap.add_argument("--vel", type=float, default=0.0)
    ap.add_argument("--uniform", action="store_true", help="uniform slabs instead of equal particle counts")
    ap.add_argument("--blend", type=float, default=1.0, help="PIC/FLIP blend (1 = pure FLIP)")
    ap.add_argument("--pile", type=int, default=0, help="extra particles packed around the first one (a cell past the P2G form switch)")
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
    if a.pile:
        pos = np.concatenate([pos, np.round(pos[0]) + np.random.default_rng(5).uniform(-0.4, 0.4, size=(a.pile, 3))])
    vel = None
    if a.vel:
        vel = np.random.default_rng(1).standard_normal(pos.shape) * a.vel
    if a.uniform:
        bounds = [round(a.n * r / world) for r in range(world + 1)]
    else:
        bounds = fd.partition_by_count(a.n, pos, world)
    if a.mode == "rccl":
        comm = fd.RcclComm()
    else:
        comm = fd.TorchComm(mode=a.mode, device=torch.device("cuda", dev))
    sim = fd.DistFluidSim(a.n, bounds, comm, device=dev, flip_blend=a.blend)
    sim.upload_global(pos, vel)
    stats = []
    for _ in range(a.steps):
        stats.append(sim.step())
    p, v, ids = sim.download_local()
    idx = sim.field(fs.FIELD.INDICES)[sim.xs:sim.xe]
    cont = sim.field(fs.FIELD.CONTAINER)[sim.xs:sim.xe]
    pres = sim.field(fs.FIELD.PRESSURE)[sim.xs:sim.xe]
    vx = sim.field(fs.FIELD.VEL)[:, sim.xs:sim.xe]
    gathered = [None] * world
    dist.gather_object((p, v, ids, idx, cont, pres, vx, bounds, comm.calls), gathered if rank == 0 else None, dst=0)
    if rank == 0:
        P = np.concatenate([g[0] for g in gathered]); V = np.concatenate([g[1] for g in gathered]); I = np.concatenate([g[2] for g in gathered])
        o = np.argsort(I)
        np.savez(a.out, pos=P[o], vel=V[o], ids=I[o], indices=np.concatenate([g[3] for g in gathered]),
                 container=np.concatenate([g[4] for g in gathered]), pressure=np.concatenate([g[5] for g in gathered]),
                 velgrid=np.concatenate([g[6] for g in gathered], axis=1), bounds=np.array(bounds),
                 num_active=np.array([s["num_active"] for s in stats]), outer=np.array([s["outer_passes"] for s in stats]),
                 iters=np.array([s["cg_iters"] for s in stats]), dt=np.array([s["dt_out"] for s in stats]),
                 counts=np.array([len(g[2]) for g in gathered]), calls=np.array([gathered[0][8]["sendrecv"], gathered[0][8]["allreduce"]]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
