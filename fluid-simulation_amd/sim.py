"""FluidSim: the reference's step loop surface (fluid.cc:1368-1507) over the C ABI."""
import ctypes as C
import numpy as np

from ._lib import lib, check, Params, StepStats, FIELD

_FIELD_DTYPE = {
    FIELD.CONTAINER: (np.float32, 1), FIELD.WEIGHTS: (np.float32, 1), FIELD.OUTPUT: (np.float32, 1),
    FIELD.VEL: (np.float64, 3), FIELD.VEL_BEFORE: (np.float64, 3), FIELD.INDICES: (np.int32, 1),
    FIELD.RHS: (np.float32, 1), FIELD.DIVER: (np.float32, 1), FIELD.DIVER2: (np.float32, 1),
    FIELD.PRESSURE: (np.float64, 1), FIELD.SOLID: (np.uint8, 1), FIELD.FLAGS: (np.uint8, 1),
}


def grid_bounds(n):
    """(lo, hi) cell coordinates for n cells per axis; n=121 -> (-60, 60) like fluid.cc:1159."""
    lo = -(n // 2)
    return lo, lo + n - 1


_VDB_COMPRESSION = {"zip": 3, "active_mask": 2}


def write_vdb(path, grids, compression="zip"):
    """Write dense float32 (n,n,n) arrays as the unnamed FloatGrids of fluid.cc:1161-1164,1503 (OpenVDB file format 224);
    compression "zip" = ZIP | ACTIVE_MASK (the library's default), "active_mask" = ACTIVE_MASK only."""
    import ctypes as C
    if isinstance(grids, np.ndarray) and grids.ndim == 3:
        grids = [grids]
    arrs = [np.ascontiguousarray(g, dtype=np.float32) for g in grids]
    n = arrs[0].shape[0]
    assert all(a.shape == (n, n, n) for a in arrs)
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    check(lib.fluid_write_vdb_ex(str(path).encode(), n, len(arrs), ptrs, _VDB_COMPRESSION[compression]))


class VdbStream:
    """The final mygrids.vdb of the reference (every step's grid, fluid.cc:1366,1450,1508) written one grid at a time."""

    def __init__(self, path, n, n_grids, compression="zip"):
        import ctypes as C
        self._h = C.c_void_p()
        self.n = n
        check(lib.fluid_vdb_open(str(path).encode(), n, n_grids, _VDB_COMPRESSION[compression], C.byref(self._h)))

    def append(self, grid):
        a = np.ascontiguousarray(grid, dtype=np.float32)
        assert a.shape == (self.n,) * 3
        check(lib.fluid_vdb_append(self._h, a.ctypes.data))

    def close(self):
        h, self._h = self._h, None
        if h:
            check(lib.fluid_vdb_close(h))


def water_cube_drop(n, ppc, seed=0):
    """Synthetic input of SURVEY.md 8(d) (generalises fluid.cc:1176,1349): (npart,3) float64 positions."""
    cnt = lib.fluid_scene_water_cube_drop(n, ppc, seed, None)
    if cnt < 0:
        raise ValueError("bad scene arguments")
    pos = np.empty((cnt, 3), dtype=np.float64)
    got = lib.fluid_scene_water_cube_drop(n, ppc, seed, pos.ctypes.data_as(C.c_void_p))
    assert got == cnt
    return pos


def reference_scatter(lo=-20, hi=20, points_per_volume=10.0, seed=0, boundary=60):
    """The reference's own initial particles (fluid.cc:1176,1347-1350: fill(CoordBBox(lo,hi)) + UniformPointScatter with
    std::mt19937(seed)), restated on the host (csrc/scene_scatter.cpp): (npart,3) float64 positions.  The defaults are the
    reference's scene on its 121^3 grid: 689210 particles."""
    l3 = (C.c_int32 * 3)(*([lo] * 3 if np.isscalar(lo) else lo))
    h3 = (C.c_int32 * 3)(*([hi] * 3 if np.isscalar(hi) else hi))
    cnt = lib.fluid_scene_uniform_scatter(l3, h3, points_per_volume, seed, boundary, None)
    if cnt < 0:
        raise ValueError("bad scene arguments")
    pos = np.empty((cnt, 3), dtype=np.float64)
    got = lib.fluid_scene_uniform_scatter(l3, h3, points_per_volume, seed, boundary, pos.ctypes.data_as(C.c_void_p))
    assert got == cnt
    return pos


class FluidSim:
    """One simulation on one MI355X.  Mirrors what main() owns in the reference:
    grids + PointList + dt, and one ``step()`` = one iteration of fluid.cc:1378-1490."""

    def __init__(self, n=121, device=0, precision="fp64", **kw):
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
        self._h = C.c_void_p()
        check(lib.fluid_create(C.byref(p), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.fluid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene -------------------------------------------------------------------------
    def set_solid(self, solid):
        s = np.ascontiguousarray(solid, dtype=np.uint8).reshape(-1)
        assert s.size == self.n ** 3
        check(lib.fluid_set_solid(self._h, s.ctypes.data_as(C.c_void_p)))

    def upload_particles(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        v = None
        if vel is not None:
            vel = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 3)
            assert vel.shape == pos.shape
            v = vel.ctypes.data_as(C.c_void_p)
        check(lib.fluid_upload_particles(self._h, pos.shape[0], pos.ctypes.data_as(C.c_void_p), v))

    def download_particles(self):
        n = lib.fluid_num_particles(self._h)
        pos = np.empty((n, 3), dtype=np.float64)
        vel = np.empty((n, 3), dtype=np.float64)
        check(lib.fluid_download_particles(self._h, pos.ctypes.data_as(C.c_void_p), vel.ctypes.data_as(C.c_void_p)))
        return pos, vel

    @property
    def num_particles(self):
        return lib.fluid_num_particles(self._h)

    @property
    def dt(self):
        d = C.c_double()
        check(lib.fluid_get_dt(self._h, C.byref(d)))
        return d.value

    @dt.setter
    def dt(self, v):
        check(lib.fluid_set_dt(self._h, float(v)))

    # ---- step + phases -----------------------------------------------------------------
    def step(self):
        st = StepStats()
        check(lib.fluid_step(self._h, C.byref(st)))
        return st.as_dict()

    def p2g(self):
        check(lib.fluid_p2g(self._h))

    def flags_index(self):
        check(lib.fluid_flags_index(self._h))

    def rhs_div(self, which=0):
        check(lib.fluid_rhs_div(self._h, which))

    def solve(self):
        check(lib.fluid_solve(self._h))

    def vel_update(self):
        check(lib.fluid_vel_update(self._h))

    def pressure_pass(self):
        e = C.c_double()
        check(lib.fluid_pressure_pass(self._h, C.byref(e)))
        return e.value

    def flip_advect(self):
        check(lib.fluid_flip_advect(self._h))

    def stats(self):
        st = StepStats()
        check(lib.fluid_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    # ---- fields ------------------------------------------------------------------------
    def _solver_dtype(self):
        return np.float64 if self.precision == "fp64" else np.float32

    def field(self, fid):
        if fid in (FIELD.SEARCH, FIELD.Q):
            dt, comps = self._solver_dtype(), 1
        else:
            dt, comps = _FIELD_DTYPE[fid]
        n = self.n
        arr = np.empty((comps, n, n, n) if comps > 1 else (n, n, n), dtype=dt)
        check(lib.fluid_download_field(self._h, fid, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return arr

    def upload_field(self, fid, arr):
        if fid in (FIELD.SEARCH, FIELD.Q):
            dt = self._solver_dtype()
        else:
            dt = _FIELD_DTYPE[fid][0]
        a = np.ascontiguousarray(arr, dtype=dt)
        check(lib.fluid_upload_field(self._h, fid, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def extrapolate(self):
        """fluid.cc:705-802 after p2g(): velocities for every cell inside W (dead code in the reference; optional here).  Returns the passes run."""
        n = C.c_int32()
        check(lib.fluid_extrapolate(self._h, C.byref(n)))
        return n.value

    def droplets(self):
        """The closed pockets of the last step's pressure system that were solved apart (include/fluid_hip.h, fluid_get_droplets):
        an (n, 64) int64 array of cell indices per component, ascending, padded with -1."""
        n = C.c_int32(0)
        check(lib.fluid_get_droplets(self._h, C.byref(n), None, 0))
        cells = np.full((max(n.value, 0), 64), -1, dtype=np.int64)
        if n.value > 0:
            check(lib.fluid_get_droplets(self._h, C.byref(n), cells.ctypes.data_as(C.c_void_p), n.value))
        return cells

    def resample(self, per_cell):
        """fluid.cc:1053-1080: at most per_cell particles per base cell (upload order); returns how many were parked outside the grid."""
        n = C.c_int64()
        check(lib.fluid_resample(self._h, int(per_cell), C.byref(n)))
        return n.value

    def stencil_apply(self, reps=1, box=0):
        ms = C.c_float()
        check(lib.fluid_stencil_apply(self._h, reps, box, C.byref(ms)))
        return ms.value

    def stencil_apply_hbm(self, reps=1, box=0, footprint_bytes=1 << 30):
        """The same sweep rotating over separate copies of (s, q, flags) that together exceed `footprint_bytes`
        (well above the 256 MiB Infinity Cache): returns (ms per launch, sets used)."""
        ms, ns = C.c_float(), C.c_int32()
        check(lib.fluid_stencil_apply_hbm(self._h, reps, box, footprint_bytes, C.byref(ns), C.byref(ms)))
        return ms.value, ns.value

    # ---- profiling -----------------------------------------------------------------------
    def profile_enable(self, sample_every):
        check(lib.fluid_profile_enable(self._h, sample_every))

    def profile_reset(self):
        check(lib.fluid_profile_reset(self._h))

    def profile_read(self, klass):
        nl, ns = C.c_int64(), C.c_int64()
        ms, cells = C.c_double(), C.c_double()
        check(lib.fluid_profile_read(self._h, klass, C.byref(nl), C.byref(ns), C.byref(ms), C.byref(cells)))
        return {"launches": nl.value, "sampled": ns.value, "total_ms": ms.value, "cells": cells.value}
