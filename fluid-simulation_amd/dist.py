"""Multi-GPU plumbing: one process per GPU, transport = torch.distributed.

The decomposition, halo protocol and distributed PCG live in the C++ host
(csrc/fluid_api.hip, "Multi-GPU" section).  This module only supplies the two callbacks of
`fluid_comm_t` (include/fluid_hip.h):

  sendrecv   exchange with the x-neighbour ranks (batched isend/irecv)
  allreduce  small in-place reductions (PCG scalars, bounding box, max speed)

Modes
  "device"  backend nccl (= RCCL over xGMI): the pointers are wrapped zero-copy as CUDA tensors and
            the collectives are enqueued on the library's own HIP stream (no host sync).
  "staged"  any backend (gloo in the tests): device -> pinned host -> collective -> device.
  "host"    the pointers are host memory (CPU-only tests of the protocol, no GPU involved).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import lib, check, Params, StepStats
from .sim import FluidSim, grid_bounds

SENDRECV_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                         C.c_void_p, C.c_size_t, C.c_void_p)
ALLREDUCE_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p)


class FluidComm(C.Structure):
    """fluid_comm_t"""
    _fields_ = [("rank", C.c_int32), ("size", C.c_int32), ("ctx", C.c_void_p), ("sendrecv", SENDRECV_T), ("allreduce", ALLREDUCE_T)]


lib.fluid_create_dist.restype = C.c_int
lib.fluid_create_dist.argtypes = [C.POINTER(Params), C.POINTER(FluidComm), C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
lib.fluid_upload_particles_ids.restype = C.c_int
lib.fluid_upload_particles_ids.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
lib.fluid_download_particles_ids.restype = C.c_int64
lib.fluid_download_particles_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.fluid_partition_by_count.restype = C.c_int
lib.fluid_partition_by_count.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]

lib.fluid_rccl_unique_id.restype = C.c_int
lib.fluid_rccl_unique_id.argtypes = [C.c_char_p, C.c_void_p]
lib.fluid_rccl_comm_create.restype = C.c_int
lib.fluid_rccl_comm_create.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(FluidComm)]
lib.fluid_rccl_comm_destroy.restype = C.c_int
lib.fluid_rccl_comm_destroy.argtypes = [C.POINTER(FluidComm)]
lib.fluid_rccl_last_error.restype = C.c_char_p

_DT = {0: torch.float64, 1: torch.int32, 2: torch.int64}
_OP = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}
_ESZ = {0: 8, 1: 4, 2: 8}


class _DevPtr:
    """Zero-copy view of raw device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def partition_by_count(n, pos, size):
    """Equal-particle-count split of the x planes (host only): bounds[size+1]."""
    pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
    b = (C.c_int32 * (size + 1))()
    check(lib.fluid_partition_by_count(n, pos.shape[0], pos.ctypes.data_as(C.c_void_p), size, b))
    return list(b)


class TorchComm:
    """The two fluid_comm_t callbacks over a torch.distributed process group."""

    def __init__(self, mode="device", group=None, device=None):
        assert mode in ("device", "staged", "host")
        self.mode = mode
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.device = device
        self.error = None
        self._sr = SENDRECV_T(self._sendrecv)
        self._ar = ALLREDUCE_T(self._allreduce)
        self.struct = FluidComm(self.rank, self.size, None, self._sr, self._ar)
        self.calls = {"sendrecv": 0, "allreduce": 0}

    # -- pointer -> tensor ------------------------------------------------------------------
    def _bytes(self, ptr, n):
        if self.mode == "host":
            return torch.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=torch.uint8)
        return torch.as_tensor(_DevPtr(ptr, n), device=self.device)

    def _stream_ctx(self, stream):
        if self.mode == "device" and stream:
            return torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device))
        import contextlib
        return contextlib.nullcontext()

    def _global(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    # -- callbacks ----------------------------------------------------------------------------
    def _sendrecv(self, ctx, send_lo, nlo_s, recv_lo, nlo_r, send_hi, nhi_s, recv_hi, nhi_r, stream):
        try:
            self.calls["sendrecv"] += 1
            lo, hi = self.rank - 1, self.rank + 1
            staged = self.mode == "staged"
            if staged:
                torch.cuda.synchronize(self.device)
            keep, ops, back = [], [], []
            with self._stream_ctx(stream):
                def tx(ptr, n, peer):
                    t = self._bytes(ptr, n)
                    if staged:
                        t = t.cpu()
                    keep.append(t)
                    ops.append(dist.P2POp(dist.isend, t, self._global(peer), self.group))

                def rx(ptr, n, peer):
                    t = self._bytes(ptr, n)
                    if staged:
                        h = torch.empty(n, dtype=torch.uint8)
                        back.append((t, h))
                        t = h
                    keep.append(t)
                    ops.append(dist.P2POp(dist.irecv, t, self._global(peer), self.group))

                # receives first, then sends; the order is the same on every rank
                if nlo_r:
                    rx(recv_lo, nlo_r, lo)
                if nhi_r:
                    rx(recv_hi, nhi_r, hi)
                if nlo_s:
                    tx(send_lo, nlo_s, lo)
                if nhi_s:
                    tx(send_hi, nhi_s, hi)
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
            self.calls["allreduce"] += 1
            n = count * _ESZ[dtype]
            staged = self.mode == "staged"
            if staged:
                torch.cuda.synchronize(self.device)
            with self._stream_ctx(stream):
                t = self._bytes(buf, n).view(_DT[dtype])
                if staged:
                    h = t.cpu()
                    dist.all_reduce(h, op=_OP[op], group=self.group)
                    t.copy_(h)
                else:
                    dist.all_reduce(t, op=_OP[op], group=self.group)
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
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.error = None
        self.calls = {"sendrecv": -1, "allreduce": -1}
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


class DistFluidSim(FluidSim):
    """One rank of an x-slab decomposed simulation.  Same step() surface as FluidSim."""

    def __init__(self, n, bounds, comm, device=0, precision="fp64", **kw):
        p = Params()
        check(lib.fluid_default_params(C.byref(p)))
        p.n = n
        p.device = device
        p.precision = {"fp64": 0, "fp32": 1}[precision]
        for k, v in kw.items():
            if k == "gravity":
                p.gravity[0], p.gravity[1], p.gravity[2] = v
            elif k == "preconditioner":
                p.reserved = {"mg": 0, "jacobi": 1}[v]
            elif hasattr(p, k):
                setattr(p, k, v)
            else:
                raise TypeError(f"unknown parameter {k}")
        self.params = p
        self.n = n
        self.precision = precision
        self.lo, self.hi = grid_bounds(n)
        self.comm = comm
        self.bounds = list(bounds)
        self.xs, self.xe = self.bounds[comm.rank], self.bounds[comm.rank + 1]
        b = (C.c_int32 * len(self.bounds))(*self.bounds)
        self._h = C.c_void_p()
        check(lib.fluid_create_dist(C.byref(p), C.byref(comm.struct), b, C.byref(self._h)))

    def _check(self, rc):
        if rc != 0 and self.comm.error is not None:
            raise self.comm.error
        check(rc)

    def upload_global(self, pos, vel=None):
        """Every rank passes the SAME global arrays; each keeps the particles whose base cell x is in its slab.
        Global id = index in the global array."""
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        bx = np.floor(np.abs(pos[:, 0]) + 0.5) * np.sign(pos[:, 0])  # C round(): half away from zero
        bx = bx.astype(np.int64) - self.lo
        lo = self.xs if self.comm.rank > 0 else -(1 << 60)
        hi = self.xe if self.comm.rank < self.comm.size - 1 else (1 << 60)
        sel = np.nonzero((bx >= lo) & (bx < hi))[0]
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

    def step(self):
        st = StepStats()
        self._check(lib.fluid_step(self._h, C.byref(st)))
        return st.as_dict()
