"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fluid_oracle.cpp header).

ctypes wrapper of the CPU restatement (liboracle.so) and, when present, of the reference's
own vendored-Eigen solver build (oracle/_ref/libeigen_ref.so).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libeigen_ref.so")


def build(ref=True):
    """Compile the restatement (always) and the Eigen reference build (only where
    /root/reference exists, i.e. in the build container)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference/Eigen"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


if not os.path.exists(_SO):
    build(ref=False)
_lib = C.CDLL(_SO)
_P = C.c_void_p
_lib.oracle_create.restype = _P
_lib.oracle_create.argtypes = [C.c_int, C.c_double, C.c_double, _P, C.c_double, C.c_double, C.c_double]
_lib.oracle_destroy.argtypes = [_P]
_lib.oracle_set_ref_solver.argtypes = [_P, _P]
_lib.oracle_set_cg_tol.argtypes = [_P, C.c_double]
_lib.oracle_set_threads.argtypes = [_P, C.c_int]
_lib.oracle_set_flip_blend.argtypes = [_P, C.c_double]
_lib.oracle_set_dt.argtypes = [_P, C.c_double]
_lib.oracle_get_dt.restype = C.c_double
_lib.oracle_get_dt.argtypes = [_P]
_lib.oracle_set_solid.restype = C.c_int
_lib.oracle_set_solid.argtypes = [_P, _P]
_lib.oracle_set_particles.argtypes = [_P, C.c_long, _P, _P]
_lib.oracle_num_particles.restype = C.c_long
_lib.oracle_num_particles.argtypes = [_P]
_lib.oracle_get_particles.argtypes = [_P, _P, _P]
for _n in ("p2g", "flags_index", "rhs_div", "build_matrix", "solve", "vel_update", "flip_advect"):
    getattr(_lib, "oracle_" + _n).argtypes = [_P]
_lib.oracle_pressure_pass.restype = C.c_double
_lib.oracle_pressure_pass.argtypes = [_P]
_lib.oracle_step.argtypes = [_P, C.c_int]
_lib.oracle_extrapolate.argtypes = [_P]
_lib.oracle_resample.argtypes = [_P, C.c_int]
_lib.oracle_stats.argtypes = [_P, _P]
_lib.oracle_get_field.restype = C.c_int
_lib.oracle_get_field.argtypes = [_P, C.c_int, _P]
_lib.oracle_num_active.restype = C.c_int
_lib.oracle_num_active.argtypes = [_P]
_lib.oracle_get_b.argtypes = [_P, _P, _P, _P]
_lib.oracle_get_triplets.restype = C.c_int
_lib.oracle_get_triplets.argtypes = [_P, _P, _P, _P]
_lib.oracle_spline.restype = C.c_double
_lib.oracle_spline.argtypes = [C.c_double]
_lib.oracle_cg_triplets.argtypes = [C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_double, _P, _P]

_ref = None


def ref_lib():
    """The vendored-Eigen build of the reference's solver, or None when it is not there."""
    global _ref
    if _ref is None and os.path.exists(_REF_SO):
        _ref = C.CDLL(_REF_SO)
        for n in ("eigen_ref_icpcg", "eigen_ref_jacobi_cg"):
            getattr(_ref, n).argtypes = [C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P]
        _ref.eigen_ref_version.restype = C.c_char_p
    return _ref


def _ptr(a):
    return a.ctypes.data_as(_P)


_SPLINE_SO = os.path.join(_HERE, "_ref", "libspline_ref.so")


def ref_spline(x):
    """The reference's own spline() (fluid.cc:22-37 compiled as it is, oracle/_ref/libspline_ref.so) on an array; None if
    the reference build is not present."""
    if not os.path.exists(_SPLINE_SO):
        return None
    lib = C.CDLL(_SPLINE_SO)
    lib.ref_spline_n.argtypes = [C.c_long, _P, _P]
    x = np.ascontiguousarray(x, dtype=np.float64)
    w = np.empty_like(x)
    lib.ref_spline_n(x.size, x.ctypes.data_as(_P), w.ctypes.data_as(_P))
    return w


def spline(x):
    return _lib.oracle_spline(float(x))


def _solve_triplets(fn, n, rows, cols, vals, b):
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n)
    it = C.c_int()
    err = C.c_double()
    fn(n, len(vals), _ptr(rows), _ptr(cols), _ptr(vals), _ptr(b), _ptr(x), C.byref(it), C.byref(err))
    return x, it.value, err.value


def cg_triplets(n, rows, cols, vals, b, tol=2.220446049250313e-16):
    """The restated Jacobi-CG on a small SPD system given as triplets."""
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n)
    it = C.c_int()
    err = C.c_double()
    _lib.oracle_cg_triplets(n, len(vals), _ptr(rows), _ptr(cols), _ptr(vals), _ptr(b), _ptr(x), tol, C.byref(it), C.byref(err))
    return x, it.value, err.value


def eigen_icpcg(n, rows, cols, vals, b):
    return _solve_triplets(ref_lib().eigen_ref_icpcg, n, rows, cols, vals, b)


def eigen_jacobi_cg(n, rows, cols, vals, b):
    return _solve_triplets(ref_lib().eigen_ref_jacobi_cg, n, rows, cols, vals, b)


FIELD_DTYPE = {0: np.float32, 1: np.float32, 2: np.float64, 3: np.float64, 4: np.int32, 5: np.float32, 6: np.float32,
               7: np.float64, 8: np.float32, 9: np.uint8, 10: np.float32, 11: np.float32, 12: np.float32, 13: np.float32}


class Oracle:
    """CPU restatement of the reference step; same phase names as the C ABI."""

    def __init__(self, n=121, dx=1.0, rho=1.0, gravity=(0.0, -10.0, 0.0), max_dt=0.1, outer_tol=0.1, update_frac=0.1,
                 use_ref_solver=False):
        g = (C.c_double * 3)(*gravity)
        self.n = n
        self._h = _lib.oracle_create(n, dx, rho, g, max_dt, outer_tol, update_frac)
        self.uses_ref_solver = False
        if use_ref_solver:
            r = ref_lib()
            if r is None:
                raise RuntimeError("oracle/_ref/libeigen_ref.so not built")
            _lib.oracle_set_ref_solver(self._h, C.cast(r.eigen_ref_icpcg, _P))
            self.uses_ref_solver = True

    def close(self):
        if self._h:
            _lib.oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_solid(self, solid):
        s = np.ascontiguousarray(solid, dtype=np.uint8).reshape(-1)
        if _lib.oracle_set_solid(self._h, _ptr(s)) != 0:
            raise ValueError("cells outside W must be solid")

    def set_particles(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        vel = np.zeros_like(pos) if vel is None else np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 3)
        _lib.oracle_set_particles(self._h, pos.shape[0], _ptr(pos), _ptr(vel))

    def particles(self):
        n = _lib.oracle_num_particles(self._h)
        pos = np.empty((n, 3))
        vel = np.empty((n, 3))
        _lib.oracle_get_particles(self._h, _ptr(pos), _ptr(vel))
        return pos, vel

    @property
    def dt(self):
        return _lib.oracle_get_dt(self._h)

    @dt.setter
    def dt(self, v):
        _lib.oracle_set_dt(self._h, float(v))

    def set_cg_tol(self, tol):
        _lib.oracle_set_cg_tol(self._h, tol)

    def set_threads(self, n):
        """Host threads of the particle loops (1 = the serial, reproducible order of the parity tests; > 1 mirrors the
        reference's tbb::parallel_for with per-cell locks: float32 sums then depend on the schedule, as in the reference)."""
        _lib.oracle_set_threads(self._h, int(n))

    def set_flip_blend(self, b):
        """1 = the reference's pure FLIP; < 1 = PIC/FLIP blend (build extension, BASELINE config 1)."""
        _lib.oracle_set_flip_blend(self._h, float(b))

    def p2g(self): _lib.oracle_p2g(self._h)
    def flags_index(self): _lib.oracle_flags_index(self._h)
    def rhs_div(self): _lib.oracle_rhs_div(self._h)
    def build_matrix(self): _lib.oracle_build_matrix(self._h)
    def solve(self): _lib.oracle_solve(self._h)
    def vel_update(self): _lib.oracle_vel_update(self._h)
    def pressure_pass(self): return _lib.oracle_pressure_pass(self._h)
    def flip_advect(self): _lib.oracle_flip_advect(self._h)

    def extrapolate(self):
        """fluid.cc:705-802 on the velocity grid as P2Gtransfer leaves it (call after p2g())."""
        _lib.oracle_extrapolate(self._h)

    def resample(self, per_cell):
        """fluid.cc:1053-1080: park the particles beyond `per_cell` per base cell (index order)."""
        _lib.oracle_resample(self._h, int(per_cell))

    def step(self, max_passes=0):
        _lib.oracle_step(self._h, max_passes)
        return self.stats()

    def stats(self):
        a = np.zeros(8)
        _lib.oracle_stats(self._h, _ptr(a))
        return {"dt_out": a[0], "num_active": int(a[1]), "outer_passes": int(a[2]), "cg_iters": int(a[3]),
                "cg_iters_last": int(a[4]), "relres": a[5], "error": a[6], "max_speed": a[7]}

    def field(self, fid):
        n = self.n
        dt = FIELD_DTYPE[fid]
        arr = np.empty((3, n, n, n) if fid in (2, 3) else (n, n, n), dtype=dt)
        if _lib.oracle_get_field(self._h, fid, _ptr(arr)) != 0:
            raise ValueError("unknown field")
        return arr

    def system(self):
        """(rows, cols, vals, b, b2, p) of the last pass in index space."""
        na = _lib.oracle_num_active(self._h)
        nnz = _lib.oracle_get_triplets(self._h, None, None, None)
        rows = np.empty(nnz, np.int32)
        cols = np.empty(nnz, np.int32)
        vals = np.empty(nnz)
        _lib.oracle_get_triplets(self._h, _ptr(rows), _ptr(cols), _ptr(vals))
        b = np.zeros(na)
        b2 = np.zeros(na)
        p = np.zeros(na)
        _lib.oracle_get_b(self._h, _ptr(b), _ptr(b2), _ptr(p))
        return rows, cols, vals, b, b2, p
