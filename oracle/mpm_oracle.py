"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/mpm_oracle.cpp header).

ctypes wrapper of the CPU restatement of the snow-MPM step (libmpm_oracle.so) and, when present, of the reference's own
constitutive functions (oracle/_ref/libmpm_ref.so, built from deformHeader.h / mpm.cc against the vendored Eigen).
Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmpm_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libmpm_ref.so")
_EIGEN_SO = os.path.join(_HERE, "_ref", "libeigen_ref.so")

if not os.path.exists(_SO):
    subprocess.check_call(["make", "-s", "-C", _HERE, "libmpm_oracle.so"])
_lib = C.CDLL(_SO)
_P = C.c_void_p
_D = C.c_double


class Params(C.Structure):
    _fields_ = [("E", _D), ("nu", _D), ("beta", _D), ("epsilon", _D), ("thetac", _D), ("thetas", _D), ("max_dt", _D),
                ("dx", _D), ("gravity", _D * 3)]


class Stats(C.Structure):
    _fields_ = [("dt_in", _D), ("dt_out", _D), ("cg_error", _D), ("max_speed", _D), ("max_grad", _D), ("max_fp", _D),
                ("max_fe", _D), ("max_force", _D * 3), ("max_mi", _D), ("max_force_coeff2", _D),
                ("num_active", C.c_int32), ("cg_iters", C.c_int32), ("any_active", C.c_int32), ("pad_", C.c_int32)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("max_force", "pad_")}
        d["max_force"] = list(self.max_force)
        return d


_lib.mpm_oracle_create.restype = _P
_lib.mpm_oracle_create.argtypes = [C.c_int, C.c_int]
_lib.mpm_oracle_destroy.argtypes = [_P]
_lib.mpm_oracle_set_ref_solver.argtypes = [_P, _P]
_lib.mpm_oracle_set_particles.restype = C.c_long
_lib.mpm_oracle_set_particles.argtypes = [_P, C.c_long, _P, _P]
_lib.mpm_oracle_set_state.argtypes = [_P, _P, _P, _P, C.c_int]
_lib.mpm_oracle_set_dt.argtypes = [_P, _D]
_lib.mpm_oracle_set_transposed.argtypes = [_P, C.c_int]
_lib.mpm_oracle_get_dt.restype = _D
_lib.mpm_oracle_get_dt.argtypes = [_P]
_lib.mpm_oracle_num_particles.restype = C.c_long
_lib.mpm_oracle_num_particles.argtypes = [_P]
_lib.mpm_oracle_step.argtypes = [_P, C.POINTER(Params), C.POINTER(Stats)]
_lib.mpm_oracle_get_particles.argtypes = [_P, C.c_int, _P]
_lib.mpm_oracle_get_field.argtypes = [_P, C.c_int, _P]
_lib.mpm_oracle_system_size.restype = C.c_long
_lib.mpm_oracle_system_size.argtypes = [_P, C.POINTER(C.c_long)]
_lib.mpm_oracle_get_system.argtypes = [_P, _P, _P, _P, _P, _P]
for _n in ("spline", "spline2", "spline_gradient"):
    getattr(_lib, "mpm_oracle_" + _n).restype = _D
    getattr(_lib, "mpm_oracle_" + _n).argtypes = [_D]


def _ptr(a):
    return a.ctypes.data_as(_P)


def _bind_functions(lib, prefix):
    for n in ("spline", "spline2", "spline_gradient"):
        getattr(lib, prefix + n).restype = _D
        getattr(lib, prefix + n).argtypes = [_D]
    getattr(lib, prefix + "getR").argtypes = [_P, _P]
    getattr(lib, prefix + "getS").argtypes = [_P, _P]
    getattr(lib, prefix + "getSigma").argtypes = [_D, _D, _D, _P, _P, _P]
    getattr(lib, prefix + "dPsydFdF").argtypes = [_P, _P, _D, _D, C.c_int, _P]
    getattr(lib, prefix + "clamp").argtypes = [_P, _P, _D, _D, _P, _P]


_bind_functions(_lib, "mpm_oracle_")


class Functions:
    """The constitutive functions of one library: the restatement (prefix mpm_oracle_) or the reference's own code
    (oracle/_ref/libmpm_ref.so, prefix mpm_ref_)."""

    def __init__(self, lib, prefix):
        self._l, self._p = lib, prefix

    def _f(self, n):
        return getattr(self._l, self._p + n)

    def spline(self, x): return self._f("spline")(x)
    def spline2(self, x): return self._f("spline2")(x)
    def spline_gradient(self, x): return self._f("spline_gradient")(x)

    def _m(self, name, F):
        F = np.ascontiguousarray(F, np.float64)
        out = np.empty((3, 3))
        self._f(name)(_ptr(F), _ptr(out))
        return out

    def getR(self, F): return self._m("getR", F)
    def getS(self, F): return self._m("getS", F)

    def getSigma(self, mu0, lambda0, eps, FE, FP):
        FE, FP = np.ascontiguousarray(FE, np.float64), np.ascontiguousarray(FP, np.float64)
        out = np.empty((3, 3))
        self._f("getSigma")(mu0, lambda0, eps, _ptr(FE), _ptr(FP), _ptr(out))
        return out

    def dPsydFdF(self, gradW, F, lam, mu, i):
        g, F = np.ascontiguousarray(gradW, np.float64), np.ascontiguousarray(F, np.float64)
        out = np.empty((3, 3))
        self._f("dPsydFdF")(_ptr(g), _ptr(F), lam, mu, i, _ptr(out))
        return out

    def clamp(self, tFE, FP, minv, maxv):
        tFE, FP = np.ascontiguousarray(tFE, np.float64), np.ascontiguousarray(FP, np.float64)
        a, b = np.empty((3, 3)), np.empty((3, 3))
        self._f("clamp")(_ptr(tFE), _ptr(FP), minv, maxv, _ptr(a), _ptr(b))
        return a, b


restated = Functions(_lib, "mpm_oracle_")


def reference_functions():
    """The reference's own functions, or None where oracle/_ref/libmpm_ref.so was not built (no /root/reference)."""
    if not os.path.exists(_REF_SO):
        return None
    lib = C.CDLL(_REF_SO)
    _bind_functions(lib, "mpm_ref_")
    return Functions(lib, "mpm_ref_")


def eigen_solver_pointer():
    """Address of eigen_ref_icpcg (the reference's solver object, mpm.cc:1283), or None."""
    if not os.path.exists(_EIGEN_SO):
        return None
    lib = C.CDLL(_EIGEN_SO)
    return C.cast(lib.eigen_ref_icpcg, _P), lib


class MpmOracle:
    def __init__(self, B=15, W=13, E=48000.0, nu=0.47, beta=0.5, epsilon=10.0, thetac=0.025, thetas=0.0075, max_dt=0.001,
                 dx=1.0, gravity=(0.0, -10.0, 0.0), dt0=0.001):
        self.B, self.W, self.N = B, W, 2 * B + 1
        self._h = _lib.mpm_oracle_create(B, W)
        self.params = Params(E, nu, beta, epsilon, thetac, thetas, max_dt, dx, (_D * 3)(*gravity))
        _lib.mpm_oracle_set_dt(self._h, dt0)
        self._keep = None

    def close(self):
        if self._h:
            _lib.mpm_oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_reference_solver(self):
        r = eigen_solver_pointer()
        if r is None:
            return False
        self._keep = r[1]
        _lib.mpm_oracle_set_ref_solver(self._h, r[0])
        return True

    def set_transposed(self, t):
        """1 (default): solve A^T x = b as the reference's Eigen object does; 0: A x = b (restated loop only)."""
        _lib.mpm_oracle_set_transposed(self._h, int(t))

    def set_particles(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 3)
        if vel is None:
            vel = np.tile(np.array([0.0, -50.0, 0.0]), (len(pos), 1))   # mpm.cc:484
        vel = np.ascontiguousarray(vel, np.float64).reshape(-1, 3)
        return _lib.mpm_oracle_set_particles(self._h, len(pos), _ptr(pos), _ptr(vel))

    def set_state(self, FE=None, FP=None, volume=None, step_no=0):
        a = [None if x is None else np.ascontiguousarray(x, np.float64) for x in (FE, FP, volume)]
        _lib.mpm_oracle_set_state(self._h, *[None if x is None else _ptr(x) for x in a], step_no)

    @property
    def num_particles(self):
        return _lib.mpm_oracle_num_particles(self._h)

    @property
    def dt(self):
        return _lib.mpm_oracle_get_dt(self._h)

    @dt.setter
    def dt(self, v):
        _lib.mpm_oracle_set_dt(self._h, float(v))

    def step(self):
        st = Stats()
        _lib.mpm_oracle_step(self._h, C.byref(self.params), C.byref(st))
        return st.as_dict()

    def particles(self, what):
        w = {0: 3, 1: 3, 2: 9, 3: 9, 4: 9, 5: 1}[what]
        n = self.num_particles
        out = np.empty((n, w) if w > 1 else (n,), np.float64)
        _lib.mpm_oracle_get_particles(self._h, what, _ptr(out))
        return out.reshape(-1, 3, 3) if w == 9 else out

    def field(self, fid):
        N = self.N
        if fid in (0, 1, 2):
            out = np.empty((N, N, N), np.float32)
        elif fid == 3:
            out = np.empty((N, N, N), np.int32)
        else:
            out = np.empty((N, N, N, 3), np.float64)
        _lib.mpm_oracle_get_field(self._h, fid, _ptr(out))
        return out

    def system(self):
        """(rows, cols, vals, b, x) of the last step's assembled system (mpm.cc:418-441)."""
        nnz = C.c_long()
        n = _lib.mpm_oracle_system_size(self._h, C.byref(nnz))
        rows, cols = np.empty(nnz.value, np.int32), np.empty(nnz.value, np.int32)
        vals, b, x = np.empty(nnz.value), np.empty(n), np.empty(n)
        _lib.mpm_oracle_get_system(self._h, _ptr(rows), _ptr(cols), _ptr(vals), _ptr(b), _ptr(x))
        return rows, cols, vals, b, x
