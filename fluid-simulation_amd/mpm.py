"""Host-side mirror of the reference's snow-MPM loop body (mpm.cc:1301-1436) over include/mpm_hip.h.

ctypes plumbing only: every call runs hand-written HIP kernels (csrc/mpm_step.hip); there is no CPU path.
"""
import ctypes as C

import numpy as np

from ._lib import lib, check, MpmParams, MpmStepStats


class MPM_P:
    POS, VEL, FE, FP, GRADV, VOLUME = range(6)
    WIDTH = {0: 3, 1: 3, 2: 9, 3: 9, 4: 9, 5: 1}


class MPM_F:
    CONTAINER, SOLID, OUTPUT, INDICES, VEL_BEFORE, FORCES, VEL = range(7)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def snow_cone(B=15, W=13, layers=4, points_per_voxel=400.0, seed=0):
    """The reference's initial particles (mpm.cc:1037-1052,1274-1278): positions, (n, 3) float64."""
    n = lib.mpm_scene_cone(B, W, layers, points_per_voxel, seed, None)
    if n < 0:
        raise ValueError("mpm_scene_cone: bad arguments")
    pos = np.empty((n, 3), np.float64)
    lib.mpm_scene_cone(B, W, layers, points_per_voxel, seed, _ptr(pos))
    return pos


def mpm_eval(what, a, b=None, p0=0.0, p1=0.0, p2=0.0):
    """mpm_eval of include/mpm_hip.h: a, b (n, 3, 3) -> (out0, out1)."""
    a = np.ascontiguousarray(a, np.float64).reshape(-1, 3, 3)
    bb = None if b is None else np.ascontiguousarray(b, np.float64).reshape(-1, 3, 3)
    o0, o1 = np.empty_like(a), np.empty_like(a)
    check(lib.mpm_eval(what, len(a), _ptr(a), None if bb is None else _ptr(bb), p0, p1, p2, _ptr(o0), _ptr(o1)))
    return o0, o1


class MpmSim:
    """One simulation = the state of mpm.cc's main(): grids over [-B, B]^3 and a PointList."""

    def __init__(self, B=15, W=13, **kw):
        p = MpmParams()
        check(lib.mpm_default_params(C.byref(p)))
        p.B, p.W = B, W
        for k, v in kw.items():
            if k == "gravity":
                p.gravity[:] = v
            elif hasattr(p, k):
                setattr(p, k, v)
            else:
                raise TypeError(f"unknown parameter {k}")
        self.params = p
        self.B, self.W, self.N = B, W, 2 * B + 1
        self._h = C.c_void_p()
        check(lib.mpm_create(C.byref(p), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib.mpm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_particles(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 3)
        v = None if vel is None else np.ascontiguousarray(vel, np.float64).reshape(-1, 3)
        kept = C.c_int64()
        check(lib.mpm_upload_particles(self._h, len(pos), _ptr(pos), None if v is None else _ptr(v), C.byref(kept)))
        return kept.value

    @property
    def num_particles(self):
        return lib.mpm_num_particles(self._h)

    @property
    def dt(self):
        return lib.mpm_get_dt(self._h)

    @dt.setter
    def dt(self, v):
        check(lib.mpm_set_dt(self._h, float(v)))

    def set_state(self, FE=None, FP=None, volume=None, step_no=0):
        a = [None if x is None else np.ascontiguousarray(x, np.float64) for x in (FE, FP, volume)]
        check(lib.mpm_set_state(self._h, *[None if x is None else _ptr(x) for x in a], step_no))

    def _run(self, fn):
        st = MpmStepStats()
        check(fn(self._h, C.byref(st)))
        d = st.as_dict()
        self._num_active = d["num_active"]   # sizes the buffers of system(): the handle's own count, not a caller's
        return d

    def step(self):
        return self._run(lib.mpm_step)

    def step_solve(self):
        return self._run(lib.mpm_step_solve)

    def step_advance(self):
        return self._run(lib.mpm_step_advance)

    def particles(self, what):
        w = MPM_P.WIDTH[what]
        out = np.empty((self.num_particles, w) if w > 1 else (self.num_particles,), np.float64)
        check(lib.mpm_download_particles(self._h, what, _ptr(out)))
        return out.reshape(-1, 3, 3) if w == 9 else out

    def field(self, fid):
        N = self.N
        if fid in (MPM_F.CONTAINER, MPM_F.SOLID, MPM_F.OUTPUT):
            out = np.empty((N, N, N), np.float32)
        elif fid == MPM_F.INDICES:
            out = np.empty((N, N, N), np.int32)
        else:
            out = np.empty((N, N, N, 3), np.float64)
        check(lib.mpm_download_field(self._h, fid, _ptr(out)))
        return out

    def system(self, num_active=None):
        """Right-hand side and solution of the last solve (3 doubles per unknown).  The buffers are sized by the handle's own
        unknown count; an explicit `num_active` must agree with it (mpm_download_system checks the count it is given against its own)."""
        mine = int(lib.mpm_num_active(self._h))   # the handle's own count sizes the buffers
        if num_active is not None and num_active != mine:
            raise ValueError(f"system(): the last solve had {mine} unknowns, not {num_active}")
        num_active = mine
        b = np.empty(3 * num_active, np.float64)
        x = np.empty(3 * num_active, np.float64)
        check(lib.mpm_download_system(self._h, _ptr(b), _ptr(x), 3 * num_active))
        return b, x

    def apply_matrix(self, v):
        v = np.ascontiguousarray(v, np.float64)
        y = np.empty_like(v)
        check(lib.mpm_apply_matrix(self._h, _ptr(v), _ptr(y)))
        return y
