#!/usr/bin/env python3
"""dims[0] x dims[1] x dims[2] blocks of ONE simulation as host threads of this process (in-process transport), all on GPU 0:
correctness against the one-GPU run at bench sizes and the number of iterations / exchanges the decomposed solve needs.
(The blocks share one GPU here, so the step times say nothing about scaling.)

  python tools/dist_local.py [n] [dims like 2x2x2] [steps] [decomposed|replicated]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package(); fd = fs.load_dist()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dims = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2x2x2").split("x")]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mode = sys.argv[4] if len(sys.argv) > 4 else "decomposed"
ppc = 8 if n <= 256 else 4
pos = fs.water_cube_drop(n, ppc, seed=0)
size = dims[0] * dims[1] * dims[2]
cuts = fd.partition_blocks(n, pos, dims)
grp = fd.LocalGroup(size)
sims = [None] * size
F = fs.FIELD

def work(r):
    sim = fd.DistFluidSim(n, dims, cuts, grp.comms[r], dist_solve=mode)
    sims[r] = sim
    sim.upload_global(pos)
    out = []
    for _ in range(steps):
        t0 = time.perf_counter(); st = sim.step(); out.append((time.perf_counter() - t0, st))
    p, v, ids = sim.download_local()
    return out, p, v, ids, sim.field(F.INDICES), sim.field(F.PRESSURE)

t0 = time.perf_counter()
res = grp.run(work)
print(f"n={n} dims={dims} cuts={cuts} particles={len(pos)} mode={mode}: {steps} steps in {time.perf_counter() - t0:.2f} s (all blocks on one GPU)")
ref = fs.FluidSim(n=n)
ref.upload_particles(pos)
rs = []
for _ in range(steps):
    t1 = time.perf_counter(); s = ref.step(); rs.append((time.perf_counter() - t1, s))
if steps <= 20:
    for i in range(steps):
        a, b = res[0][0][i], rs[i]
        print(f"  step {i}: blocks {a[0]*1e3:8.2f} ms  iters {a[1]['cg_iters']:4d} passes {a[1]['outer_passes']}  unknowns {a[1]['num_active']}   |  one GPU {b[0]*1e3:6.2f} ms iters {b[1]['cg_iters']:4d} passes {b[1]['outer_passes']} unknowns {b[1]['num_active']}")
else:   # a long run: sums per 100 steps (the two runs part ways in the splash: chaotic, compare totals)
    for i0 in range(0, steps, 100):
        A = [x[1] for x in res[0][0][i0:i0 + 100]]; B = [x[1] for x in rs[i0:i0 + 100]]
        gal = lambda S: sum(1 for s in S if s["paths"] & 128)
        drp = lambda R: sum(1 for s in R if s["paths"] & 64)
        print(f"  steps {i0:3d}-{i0 + len(A) - 1:3d}: blocks iters {sum(s['cg_iters'] for s in A):6d} passes {sum(s['outer_passes'] for s in A):4d} galerkin steps {gal(A):3d} "
              f"droplet steps (any rank) {sum(1 for k in range(len(A)) if any(r[0][i0 + k][1]['paths'] & 64 for r in res)):3d}  |  one GPU iters {sum(s['cg_iters'] for s in B):6d} "
              f"passes {sum(s['outer_passes'] for s in B):4d} galerkin steps {gal(B):3d} droplet steps {drp(B):3d}")
    ta, tb = sum(x[1]['cg_iters'] for x in res[0][0]), sum(x[1]['cg_iters'] for x in rs)
    print(f"  total iterations: blocks {ta}  one GPU {tb}  ratio {ta / tb:.3f}")
ids = np.concatenate([r[3] for r in res]); o = np.argsort(ids)
P = np.concatenate([r[1] for r in res])[o]; V = np.concatenate([r[2] for r in res])[o]
p, v = ref.download_particles()
idx = fd.assemble(n, sims, [r[4] for r in res]); pr = fd.assemble(n, sims, [r[5] for r in res])
rel = lambda x, y: np.linalg.norm((x - y).ravel()) / max(np.linalg.norm(y.ravel()), 1e-300)
print(f"  ids complete {np.array_equal(ids[o], np.arange(len(pos)))}  indices equal {np.array_equal(idx, ref.field(F.INDICES))}  pos {rel(P, p):.2e} vel {rel(V, v):.2e} pressure {rel(pr, ref.field(F.PRESSURE)):.2e}")
print(f"  particles per block {[len(r[3]) for r in res]}")
