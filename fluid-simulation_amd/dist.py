"""Multi-GPU plumbing: 3-D block decomposition, one process (or host thread) per block.

The decomposition, halo protocol and distributed PCG live in the C++ host (csrc/fluid_dist.hip).  This module only
supplies transports for `fluid_comm_t` (include/fluid_hip.h) and the ctypes surface:

  RcclComm    RCCL inside the C++ library (grouped ncclSend/ncclRecv + ncclAllReduce on the solver stream): bench.py
  TorchComm   the two callbacks over a torch.distributed process group
                "device"  backend nccl (= RCCL): device pointers wrapped zero-copy, collectives on the library's stream
                "staged"  any backend (gloo in the tests): device -> host -> collective -> device
  LocalGroup  several blocks in ONE process, one host thread each (csrc/comm_local.hip): the tests run 2 x 2 x 2 blocks
              on the one GPU of their box this way
"""
import ctypes as C
import threading

import numpy as np

from ._lib import lib, check, Params, StepStats
from .sim import FluidSim, grid_bounds

EXCHANGE_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                         C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p)
ALLREDUCE_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)


class FluidComm(C.Structure):
    """fluid_comm_t"""
    _fields_ = [("rank", C.c_int32), ("size", C.c_int32), ("ctx", C.c_void_p), ("exchange", EXCHANGE_T), ("allreduce", ALLREDUCE_T)]


class FluidDecomp(C.Structure):
    """fluid_decomp_t"""
    _fields_ = [("dims", C.c_int32 * 3), ("cuts", C.POINTER(C.c_int32) * 3)]


lib.fluid_create_dist.restype = C.c_int
lib.fluid_create_dist.argtypes = [C.POINTER(Params), C.POINTER(FluidComm), C.POINTER(FluidDecomp), C.POINTER(C.c_void_p)]
lib.fluid_dist_set_rebalance.restype = C.c_int
lib.fluid_dist_set_rebalance.argtypes = [C.c_void_p, C.c_int32, C.c_double]
lib.fluid_dist_get_cuts.restype = C.c_int
lib.fluid_dist_get_info.restype = C.c_int
lib.fluid_dist_get_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
lib.fluid_dist_get_cuts.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
lib.fluid_local_comm_create.restype = C.c_int
lib.fluid_local_comm_create.argtypes = [C.c_void_p, C.c_int32, C.POINTER(FluidComm)]
lib.fluid_rccl_unique_id.restype = C.c_int
lib.fluid_rccl_unique_id.argtypes = [C.c_char_p, C.c_void_p]
lib.fluid_rccl_comm_create.restype = C.c_int
lib.fluid_rccl_comm_create.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(FluidComm)]
lib.fluid_rccl_comm_destroy.restype = C.c_int
lib.fluid_rccl_comm_destroy.argtypes = [C.POINTER(FluidComm)]
lib.fluid_rccl_last_error.restype = C.c_char_p

_ESZ = {0: 8, 1: 4, 2: 8, 3: 4, 4: 1}


def default_dims(size):
    """Blocks per axis for `size` ranks: as cubic as the factorisation allows (8 -> 2 x 2 x 2, 4 -> 2 x 2 x 1, 6 -> 3 x 2 x 1)."""
    dims = [1, 1, 1]
    f, n = 2, size
    fac = []
    while n > 1:
        while n % f == 0:
            fac.append(f)
            n //= f
        f += 1
    for p in sorted(fac, reverse=True):
        dims[int(np.argmin(dims))] *= p
    return sorted(dims, reverse=True)


def partition_blocks(n, pos, dims):
    """Cut planes per axis (host only): about equal particle counts per slab, interior cuts multiples of 4."""
    pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
    d = (C.c_int32 * 3)(*dims)
    cuts = [(C.c_int32 * (dims[a] + 1))() for a in range(3)]
    check(lib.fluid_partition_blocks(n, pos.shape[0], pos.ctypes.data_as(C.c_void_p), d, cuts[0], cuts[1], cuts[2]))
    return [list(c) for c in cuts]


def uniform_cuts(n, dims):
    """Equal-width blocks, interior cuts rounded to multiples of 4."""
    return [[0] + [int(round(n * r / dims[a] / 4)) * 4 for r in range(1, dims[a])] + [n] for a in range(3)]


class _DevPtr:
    """Zero-copy view of raw device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class TorchComm:
    """The two fluid_comm_t callbacks over a torch.distributed process group."""

    def __init__(self, mode="device", group=None, device=None):
        import torch
        import torch.distributed as dist
        assert mode in ("device", "staged", "host")   # host: the pointers are host memory (CPU-only tests of the protocol)
        self.torch, self.dist = torch, dist
        self.mode = mode
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.device = device
        self.error = None
        self._ex = EXCHANGE_T(self._exchange)
        self._ar = ALLREDUCE_T(self._allreduce)
        self.struct = FluidComm(self.rank, self.size, None, self._ex, self._ar)
        self.calls = {"exchange": 0, "allreduce": 0}
        self._DT = {0: torch.float64, 1: torch.int32, 2: torch.int64, 3: torch.float32, 4: torch.uint8}
        self._OP = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}

    def _bytes(self, ptr, n):
        if self.mode == "host":
            return self.torch.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=self.torch.uint8)
        return self.torch.as_tensor(_DevPtr(ptr, n), device=self.device)

    def _stream_ctx(self, stream):
        if self.mode == "device" and stream:
            return self.torch.cuda.stream(self.torch.cuda.ExternalStream(stream, device=self.device))
        import contextlib
        return contextlib.nullcontext()

    def _global(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def _exchange(self, ctx, n, peer, sbuf, sbytes, rbuf, rbytes, stream):
        try:
            torch, dist = self.torch, self.dist
            self.calls["exchange"] += 1
            staged = self.mode == "staged"
            if staged:
                torch.cuda.synchronize(self.device)
            keep, ops, back = [], [], []
            with self._stream_ctx(stream):
                # receives first, then sends, peers ascending on both lists: the same order on every rank
                order = sorted(range(n), key=lambda i: peer[i])
                for i in order:
                    if rbytes[i]:
                        t = self._bytes(rbuf[i], rbytes[i])
                        if staged:
                            h = torch.empty(rbytes[i], dtype=torch.uint8)
                            back.append((t, h))
                            t = h
                        keep.append(t)
                        ops.append(dist.P2POp(dist.irecv, t, self._global(peer[i]), self.group))
                for i in order:
                    if sbytes[i]:
                        t = self._bytes(sbuf[i], sbytes[i])
                        if staged:
                            t = t.cpu()
                        keep.append(t)
                        ops.append(dist.P2POp(dist.isend, t, self._global(peer[i]), self.group))
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                for d, h in back:
                    d.copy_(h)
            if staged:
                torch.cuda.synchronize(self.device)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            self.error = e
            return 1

    def _allreduce(self, ctx, buf, count, dtype, op, stream):
        try:
            torch, dist = self.torch, self.dist
            self.calls["allreduce"] += 1
            n = count * _ESZ[dtype]
            staged = self.mode == "staged"
            if staged:
                torch.cuda.synchronize(self.device)
            with self._stream_ctx(stream):
                t = self._bytes(buf, n).view(self._DT[dtype])
                if staged:
                    h = t.cpu()
                    dist.all_reduce(h, op=self._OP[op], group=self.group)
                    t.copy_(h)
                else:
                    dist.all_reduce(t, op=self._OP[op], group=self.group)
            if staged:
                torch.cuda.synchronize(self.device)
            return 0
        except Exception as e:
            self.error = e
            return 1


class RcclComm:
    """fluid_comm_t backed by RCCL inside the C++ library (no Python in the PCG loop).

    The ncclUniqueId is made by rank 0 and broadcast over the existing torch.distributed group
    (bootstrap only); RCCL itself is the librccl.so of the running PyTorch, so the process holds
    one RCCL.  Call with the rank's GPU already current (torch.cuda.set_device)."""

    def __init__(self, group=None):
        import os
        import torch
        import torch.distributed as dist
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.error = None
        self.calls = {"exchange": -1, "allreduce": -1}
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.path = (path if os.path.exists(path) else "").encode()
        idbuf = (C.c_uint8 * 128)()
        obj = [bytes(idbuf)]
        if self.rank == 0 and lib.fluid_rccl_unique_id(self.path, idbuf) != 0:
            obj = [None]   # every rank must leave the broadcast below: the failure travels with it
        elif self.rank == 0:
            obj = [bytes(idbuf)]
        dist.broadcast_object_list(obj, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        if obj[0] is None:
            raise RuntimeError("fluid_rccl_unique_id failed on rank 0: " + lib.fluid_rccl_last_error().decode())
        self.struct = FluidComm()
        idb = (C.c_uint8 * 128).from_buffer_copy(obj[0])
        if lib.fluid_rccl_comm_create(self.path, idb, self.rank, self.size, C.byref(self.struct)) != 0:
            raise RuntimeError("fluid_rccl_comm_create: " + lib.fluid_rccl_last_error().decode())

    def close(self):
        lib.fluid_rccl_comm_destroy(C.byref(self.struct))


class LocalComm:
    """One rank of a LocalGroup."""

    def __init__(self, group, rank):
        self.rank, self.size = rank, group.size
        self.error = None
        self.calls = {"exchange": -1, "allreduce": -1}
        self.struct = FluidComm()
        check(lib.fluid_local_comm_create(group.handle, rank, C.byref(self.struct)))


class LocalGroup:
    """`size` blocks in this process, one host thread each, over the in-process transport (csrc/comm_local.hip)."""

    def __init__(self, size):
        self.size = size
        self.handle = C.c_void_p()
        check(lib.fluid_local_group_create(size, C.byref(self.handle)))
        self.comms = [LocalComm(self, r) for r in range(size)]

    def run(self, fn):
        """fn(rank) on `size` threads; returns the results by rank; the first exception is re-raised after all threads ended
        (a rank that fails wakes its peers out of the transport)."""
        out, errs = [None] * self.size, [None] * self.size

        def work(r):
            try:
                out[r] = fn(r)
            except BaseException as e:  # noqa: BLE001
                errs[r] = e
                lib.fluid_local_group_abort(self.handle)

        th = [threading.Thread(target=work, args=(r,)) for r in range(self.size)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        return out

    def close(self):
        if self.handle:
            lib.fluid_local_group_destroy(self.handle)
            self.handle = C.c_void_p()


class DistFluidSim(FluidSim):
    """One block of a decomposed simulation.  Same step() surface as FluidSim; field() returns the OWNED block."""

    def __init__(self, n, dims, cuts, comm, device=0, precision="fp64", **kw):
        p = Params()
        check(lib.fluid_default_params(C.byref(p)))
        p.n = n
        p.device = device
        p.precision = {"fp64": 0, "fp32": 1}[precision]
        for k, v in kw.items():
            if k == "gravity":
                p.gravity[0], p.gravity[1], p.gravity[2] = v
            elif k == "preconditioner":
                p.preconditioner = {"mg": 0, "jacobi": 1}[v]
            elif k == "solve_start":
                p.solve_start = {"warm": 0, "zero": 1}[v]
            elif k == "mg_precision":
                p.mg_precision = {"fp32": 0, "fp64": 1}[v]
            elif k == "dist_solve":
                p.dist_solve = {"auto": 0, "decomposed": 1, "replicated": 2}[v]
            elif hasattr(p, k):
                setattr(p, k, v)
            else:
                raise TypeError(f"unknown parameter {k}")
        self.params = p
        self.n = n
        self.precision = precision
        self.lo, self.hi = grid_bounds(n)
        self.comm = comm
        self.dims = list(dims)
        self.cuts = [list(c) for c in cuts]
        dc = FluidDecomp()
        self._cut_arrays = [(C.c_int32 * len(c))(*c) for c in self.cuts]
        for a in range(3):
            dc.dims[a] = self.dims[a]
            dc.cuts[a] = C.cast(self._cut_arrays[a], C.POINTER(C.c_int32))
        self._h = C.c_void_p()
        check(lib.fluid_create_dist(C.byref(p), C.byref(comm.struct), C.byref(dc), C.byref(self._h)))
        self.n_rebalanced = 0
        self._refresh_window()

    def _refresh_window(self):
        o, d, l, h = ((C.c_int32 * 3)() for _ in range(4))
        check(lib.fluid_window(self._h, o, d, l, h))
        self.origin, self.wdims, self.own_lo, self.own_hi = list(o), list(d), list(l), list(h)

    def set_rebalance(self, every, ratio=2.0):
        """Every `every` steps compare the blocks' particle counts and move the cut planes when the fullest block holds more than
        `ratio` x the mean (collective: the same call on every rank; 0 = never)."""
        check(lib.fluid_dist_set_rebalance(self._h, every, ratio))

    def current_cuts(self):
        arr = [(C.c_int32 * (self.dims[a] + 1))() for a in range(3)]
        nr = C.c_int32()
        check(lib.fluid_dist_get_cuts(self._h, arr[0], arr[1], arr[2], C.byref(nr)))
        return [list(c) for c in arr], nr.value

    def info(self):
        """{'overlap': 0 unchecked | 1 verified and in use | 2 failed its check and off, 'cg_form': 0 cg | 1 cgear, 'rebalances_refused': n}"""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.fluid_dist_get_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"overlap": a.value, "cg_form": b.value, "rebalances_refused": c.value}

    def _check(self, rc):
        if rc != 0 and getattr(self.comm, "error", None) is not None:
            raise self.comm.error
        check(rc)

    def owns(self, pos):
        """Mask of the particles whose base cell lies in this block (edge blocks reach to infinity on their outer sides)."""
        pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
        b = (np.floor(np.abs(pos) + 0.5) * np.sign(pos)).astype(np.int64) - self.lo   # C round(): half away from zero
        m = np.ones(len(pos), dtype=bool)
        for a in range(3):
            lo = self.own_lo[a] if self.own_lo[a] > 0 else -(1 << 60)
            hi = self.own_hi[a] if self.own_hi[a] < self.n else (1 << 60)
            m &= (b[:, a] >= lo) & (b[:, a] < hi)
        return m

    def upload_global(self, pos, vel=None):
        """Every rank passes the SAME global arrays; each keeps the particles of its block.  Global id = index in the global array."""
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        sel = np.nonzero(self.owns(pos))[0]
        ids = sel.astype(np.uint32)
        mypos = np.ascontiguousarray(pos[sel])
        myvel = None if vel is None else np.ascontiguousarray(np.asarray(vel, dtype=np.float64).reshape(-1, 3)[sel])
        self._check(lib.fluid_upload_particles_ids(self._h, len(sel), mypos.ctypes.data_as(C.c_void_p),
                                                   None if myvel is None else myvel.ctypes.data_as(C.c_void_p),
                                                   ids.ctypes.data_as(C.c_void_p)))

    def download_local(self):
        n = lib.fluid_download_particles_ids(self._h, None, None, None)
        pos = np.empty((n, 3)); vel = np.empty((n, 3)); ids = np.empty(n, dtype=np.uint32)
        got = lib.fluid_download_particles_ids(self._h, pos.ctypes.data_as(C.c_void_p), vel.ctypes.data_as(C.c_void_p),
                                               ids.ctypes.data_as(C.c_void_p))
        assert got == n
        return pos, vel, ids

    def num_live(self):
        """Particles this rank owns now (after a step: without the ghosts it served)."""
        return int(lib.fluid_download_particles_ids(self._h, None, None, None))

    def window_field(self, fid):
        """The handle's whole array (block + halo; the whole grid in replicated mode)."""
        from .sim import _FIELD_DTYPE
        dt, comps = _FIELD_DTYPE[fid]
        nx, ny, nz = self.wdims
        arr = np.empty((comps, nx, ny, nz) if comps > 1 else (nx, ny, nz), dtype=dt)
        check(lib.fluid_download_field(self._h, fid, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return arr

    def field(self, fid):
        """The owned block of a field."""
        a = self.window_field(fid)
        sl = tuple(slice(self.own_lo[k] - self.origin[k], self.own_hi[k] - self.origin[k]) for k in range(3))
        return a[(slice(None),) + sl] if a.ndim == 4 else a[sl]

    def step(self):
        st = StepStats()
        self._check(lib.fluid_step(self._h, C.byref(st)))
        if st.paths & 32:   # FLUID_PATH_DIST_REBALANCED: the cut planes moved, this handle covers another window now
            self._refresh_window()
            self.cuts, self.n_rebalanced = self.current_cuts()
        return st.as_dict()


def assemble(n, sims, blocks):
    """Global (n, n, n) (or (3, n, n, n)) array from the owned blocks of every rank."""
    first = blocks[0]
    out = np.zeros(((3, n, n, n) if first.ndim == 4 else (n, n, n)), dtype=first.dtype)
    for s, b in zip(sims, blocks):
        sl = tuple(slice(s.own_lo[k], s.own_hi[k]) for k in range(3))
        if first.ndim == 4:
            out[(slice(None),) + sl] = b
        else:
            out[sl] = b
    return out
