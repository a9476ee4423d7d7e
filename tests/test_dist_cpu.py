"""CPU (-m "not gpu"): the multi-rank protocol with world_size 2 and 3 over gloo, no GPU.

The transport callbacks of fluid_comm_t (fluid-simulation_amd/dist.py TorchComm, mode "host") are
driven as the C++ host drives them in its distributed PCG (one grouped neighbour exchange per
iteration + scalar all-reduces), on an x-slab decomposed 7-point Poisson problem in numpy, and
the result is compared with the undivided solve.  The geometry of the 3-D block decomposition
(cuts, block of a rank, default dims) is checked on the host as well."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import os, sys, ctypes as C
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import __graft_entry__ as entry
fs = entry.load_package(); fd = fs.load_dist()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
comm = fd.TorchComm(mode="host")
n = 12
rng = np.random.default_rng(0)
b_full = rng.standard_normal((n, n, n))
bounds = [round(n * r / world) for r in range(world + 1)]
xs, xe = bounds[rank], bounds[rank + 1]
nx = xe - xs
def P(a): return a.ctypes.data_as(C.c_void_p).value
def ring(s):   # s: (nx+2, n, n) with ring planes 0 and nx+1: one exchange call with both neighbours
    pb = s[0].nbytes
    peers, sb, rb = [], [], []
    if rank > 0: peers.append(rank - 1); sb.append(P(s[1])); rb.append(P(s[0]))
    if rank < world - 1: peers.append(rank + 1); sb.append(P(s[nx])); rb.append(P(s[nx + 1]))
    k = len(peers)
    rc = comm._exchange(None, k, (C.c_int32 * k)(*peers), (C.c_void_p * k)(*sb), (C.c_size_t * k)(*([pb] * k)),
                        (C.c_void_p * k)(*rb), (C.c_size_t * k)(*([pb] * k)), None)
    assert rc == 0, comm.error
def allsum(v):
    a = np.array(v, dtype=np.float64)
    assert comm._allreduce(None, P(a), a.size, 0, 0, None) == 0, comm.error
    return a
def apply(s):  # 7-point SPD operator 6.5 I - adjacency, zero outside
    ring(s)
    c = s[1:nx + 1]
    q = 6.5 * c - s[0:nx] - s[2:nx + 2]
    q[:, 1:] -= c[:, :-1]; q[:, :-1] -= c[:, 1:]
    q[:, :, 1:] -= c[:, :, :-1]; q[:, :, :-1] -= c[:, :, 1:]
    return q
b = b_full[xs:xe].copy()
x = np.zeros_like(b); r = b.copy()
s = np.zeros((nx + 2, n, n)); s[1:nx + 1] = r
rz = allsum([np.sum(r * r)])[0]
for it in range(200):
    q = apply(s)
    pq = allsum([np.sum(s[1:nx + 1] * q)])[0]
    al = rz / pq
    x += al * s[1:nx + 1]; r -= al * q
    rr = allsum([np.sum(r * r)])[0]
    if rr < 1e-24: break
    s[1:nx + 1] = r + (rr / rz) * s[1:nx + 1]; rz = rr
# other dtypes / ops used by the host: bbox MIN (int32), max speed MAX (int64)
a = np.array([rank + 5, -rank], dtype=np.int32); assert comm._allreduce(None, P(a), 2, 1, 2, None) == 0
assert list(a) == [5, -(world - 1)]
m = np.array([1000 + rank], dtype=np.int64); assert comm._allreduce(None, P(m), 1, 2, 1, None) == 0
assert m[0] == 1000 + world - 1
out = [None] * world
dist.all_gather_object(out, x)
if rank == 0:
    np.save(sys.argv[1], np.concatenate(out))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_slab_protocol_over_gloo(tmp_path, world):
    script = tmp_path / "w.py"
    script.write_text(WORKER % {"root": ROOT})
    out = str(tmp_path / "x.npy")
    port = 29700 + (os.getpid() % 1000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), out]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    x = np.load(out)
    # undivided reference
    n = 12
    b = np.random.default_rng(0).standard_normal((n, n, n))
    sp = np.pad(x, 1)
    Ax = 6.5 * x - (sp[:-2, 1:-1, 1:-1] + sp[2:, 1:-1, 1:-1] + sp[1:-1, :-2, 1:-1] + sp[1:-1, 2:, 1:-1] + sp[1:-1, 1:-1, :-2] + sp[1:-1, 1:-1, 2:])
    assert np.linalg.norm(Ax - b) / np.linalg.norm(b) < 1e-10


def test_block_partition_rules(fs):
    fd = fs.load_dist()
    assert fd.default_dims(8) == [2, 2, 2] and fd.default_dims(4) == [2, 2, 1] and fd.default_dims(2) == [2, 1, 1]
    assert fd.default_dims(6) == [3, 2, 1] and fd.default_dims(1) == [1, 1, 1]
    pos = fs.water_cube_drop(64, 4, 0)
    for dims in ([2, 1, 1], [2, 2, 2], [3, 1, 1], [1, 2, 2]):
        cuts = fd.partition_blocks(64, pos, dims)
        for a in range(3):
            c = cuts[a]
            assert c[0] == 0 and c[-1] == 64 and len(c) == dims[a] + 1
            assert all(c[i + 1] - c[i] >= 8 for i in range(dims[a])) or dims[a] == 1
            assert all(v % 4 == 0 for v in c[1:-1])
    # the cube is centred: two slabs split it near the middle
    c = fd.partition_blocks(64, pos, [2, 1, 1])[0]
    assert 28 <= c[1] <= 36
    with pytest.raises(fs.FluidError):
        fd.partition_blocks(16, pos, [4, 1, 1])   # blocks of < 8 cells
