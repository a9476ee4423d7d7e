"""ctypes binding of include/fluid_hip.h (one function per C entry point; no logic here)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfluid_hip.so")
if not os.path.exists(_SO):
    raise ImportError(
        f"{_SO} is missing: build it with `make -C {_HERE}` (hipcc --offload-arch=gfx950). "
        "This package has no CPU path.")
lib = C.CDLL(_SO)


class FluidError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfluid_hip error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """fluid_params_t (include/fluid_hip.h); defaults = the reference's literals (fluid.cc:1357-1367)."""
    _fields_ = [("n", C.c_int32), ("device", C.c_int32), ("dx", C.c_double), ("rho", C.c_double),
                ("gravity", C.c_double * 3), ("max_dt", C.c_double), ("outer_tol", C.c_double),
                ("update_frac", C.c_double), ("cg_tol", C.c_double), ("cg_max_iters", C.c_int32),
                ("max_outer_passes", C.c_int32), ("precision", C.c_int32), ("preconditioner", C.c_int32),
                ("flip_blend", C.c_double), ("solve_start", C.c_int32), ("mg_precision", C.c_int32),
                ("dist_solve", C.c_int32), ("pad_", C.c_int32)]


class StepStats(C.Structure):
    """fluid_step_stats_t."""
    _fields_ = [("dt_in", C.c_double), ("dt_out", C.c_double), ("error", C.c_double), ("max_speed", C.c_double),
                ("relres", C.c_double), ("num_active", C.c_int64), ("outer_passes", C.c_int32),
                ("cg_iters", C.c_int32), ("cg_iters_last", C.c_int32), ("box_lo", C.c_int32 * 3),
                ("box_hi", C.c_int32 * 3), ("paths", C.c_int32)]

    def as_dict(self):
        return {"dt_in": self.dt_in, "dt_out": self.dt_out, "error": self.error, "max_speed": self.max_speed,
                "relres": self.relres, "num_active": self.num_active, "outer_passes": self.outer_passes,
                "cg_iters": self.cg_iters, "cg_iters_last": self.cg_iters_last,
                "box_lo": list(self.box_lo), "box_hi": list(self.box_hi), "paths": self.paths}


class MpmParams(C.Structure):
    """mpm_params_t (include/mpm_hip.h); defaults = the literals of mpm.cc."""
    _fields_ = [("B", C.c_int32), ("W", C.c_int32), ("device", C.c_int32), ("cg_max_iters", C.c_int32),
                ("dx", C.c_double), ("gravity", C.c_double * 3), ("youngs_modulus", C.c_double),
                ("poisson_ratio", C.c_double), ("beta", C.c_double), ("hardening", C.c_double),
                ("theta_c", C.c_double), ("theta_s", C.c_double), ("max_dt", C.c_double), ("dt0", C.c_double),
                ("cg_tol", C.c_double), ("transpose_system", C.c_int32), ("pad_", C.c_int32)]


class MpmStepStats(C.Structure):
    """mpm_step_stats_t."""
    _fields_ = [("dt_in", C.c_double), ("dt_out", C.c_double), ("cg_error", C.c_double), ("max_speed", C.c_double),
                ("max_grad", C.c_double), ("max_fp", C.c_double), ("max_fe", C.c_double), ("max_force", C.c_double * 3),
                ("max_mi", C.c_double), ("max_force_coeff2", C.c_double), ("num_active", C.c_int32),
                ("cg_iters", C.c_int32), ("any_active", C.c_int32), ("cg_status", C.c_int32),
                ("ms_transfer", C.c_double), ("ms_forces", C.c_double), ("ms_solve", C.c_double),
                ("ms_deform", C.c_double), ("ms_advect", C.c_double), ("ms_apply_avg", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("max_force", "pad_")}
        d["max_force"] = list(self.max_force)
        return d


class FIELD:
    CONTAINER, WEIGHTS, VEL, VEL_BEFORE, INDICES, RHS, DIVER, PRESSURE, OUTPUT, SOLID = range(10)
    DIVER2, SEARCH, Q, FLAGS = 14, 15, 16, 17


class PROF:
    PCG_SQ, PCG_XR, P2G, G2P, SORT, SOLVE, MG_UP0 = range(7)


# every symbol include/fluid_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("fluid_default_params", C.c_int, [C.POINTER(Params)]),
    ("fluid_create", C.c_int, [C.POINTER(Params), C.POINTER(_P)]),
    ("fluid_destroy", C.c_int, [_P]),
    ("fluid_last_error", C.c_char_p, []),
    ("fluid_version", C.c_char_p, []),
    ("fluid_set_solid", C.c_int, [_P, _P]),
    ("fluid_upload_particles", C.c_int, [_P, C.c_int64, _P, _P]),
    ("fluid_download_particles", C.c_int, [_P, _P, _P]),
    ("fluid_num_particles", C.c_int64, [_P]),
    ("fluid_set_dt", C.c_int, [_P, C.c_double]),
    ("fluid_get_dt", C.c_int, [_P, C.POINTER(C.c_double)]),
    ("fluid_scene_water_cube_drop", C.c_int64, [C.c_int32, C.c_int32, C.c_uint64, _P]),
    ("fluid_scene_uniform_scatter", C.c_int64, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_float, C.c_uint32, C.c_int32, _P]),
    ("fluid_step", C.c_int, [_P, C.POINTER(StepStats)]),
    ("fluid_p2g", C.c_int, [_P]),
    ("fluid_flags_index", C.c_int, [_P]),
    ("fluid_rhs_div", C.c_int, [_P, C.c_int]),
    ("fluid_solve", C.c_int, [_P]),
    ("fluid_vel_update", C.c_int, [_P]),
    ("fluid_pressure_pass", C.c_int, [_P, C.POINTER(C.c_double)]),
    ("fluid_flip_advect", C.c_int, [_P]),
    ("fluid_get_stats", C.c_int, [_P, C.POINTER(StepStats)]),
    ("fluid_download_field", C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    ("fluid_upload_field", C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    ("fluid_stencil_apply", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    ("fluid_extrapolate", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("fluid_resample", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64)]),
    ("fluid_get_droplets", C.c_int, [_P, C.POINTER(C.c_int32), _P, C.c_int32]),
    ("fluid_stencil_apply_hbm", C.c_int, [_P, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    ("fluid_profile_enable", C.c_int, [_P, C.c_int]),
    ("fluid_profile_read", C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("fluid_profile_reset", C.c_int, [_P]),
    ("fluid_spline_eval", C.c_int, [C.c_int32, C.c_int32, C.c_int64, _P, _P]),
    ("fluid_dot_eval", C.c_int, [C.c_int32, C.c_int64, _P, _P, _P]),
    ("fluid_write_vdb", C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("fluid_write_vdb_ex", C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(_P), C.c_int32]),
    ("fluid_vdb_open", C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("fluid_vdb_append", C.c_int, [_P, _P]),
    ("fluid_vdb_close", C.c_int, [_P]),
    # the snow-MPM step (include/mpm_hip.h)
    ("mpm_default_params", C.c_int, [C.POINTER(MpmParams)]),
    ("mpm_create", C.c_int, [C.POINTER(MpmParams), C.POINTER(_P)]),
    ("mpm_destroy", C.c_int, [_P]),
    ("mpm_upload_particles", C.c_int, [_P, C.c_int64, _P, _P, C.POINTER(C.c_int64)]),
    ("mpm_num_particles", C.c_int64, [_P]),
    ("mpm_num_active", C.c_int32, [_P]),
    ("mpm_set_state", C.c_int, [_P, _P, _P, _P, C.c_int32]),
    ("mpm_set_dt", C.c_int, [_P, C.c_double]),
    ("mpm_get_dt", C.c_double, [_P]),
    ("mpm_step", C.c_int, [_P, C.POINTER(MpmStepStats)]),
    ("mpm_step_solve", C.c_int, [_P, C.POINTER(MpmStepStats)]),
    ("mpm_step_advance", C.c_int, [_P, C.POINTER(MpmStepStats)]),
    ("mpm_download_particles", C.c_int, [_P, C.c_int32, _P]),
    ("mpm_download_field", C.c_int, [_P, C.c_int32, _P]),
    ("mpm_download_system", C.c_int, [_P, _P, _P, C.c_int64]),
    ("mpm_apply_matrix", C.c_int, [_P, _P, _P]),
    ("mpm_eval", C.c_int, [C.c_int32, C.c_int64, _P, _P, C.c_double, C.c_double, C.c_double, _P, _P]),
    ("mpm_scene_cone", C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint32, _P]),
    # multi-GPU (argtypes with the comm struct are completed in dist.py)
    ("fluid_create_dist", C.c_int, None),
    ("fluid_window", C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("fluid_upload_particles_ids", C.c_int, [_P, C.c_int64, _P, _P, _P]),
    ("fluid_download_particles_ids", C.c_int64, [_P, _P, _P, _P]),
    ("fluid_partition_blocks", C.c_int, [C.c_int32, C.c_int64, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("fluid_local_group_create", C.c_int, [C.c_int32, C.POINTER(_P)]),
    ("fluid_local_group_destroy", C.c_int, [_P]),
    ("fluid_local_group_abort", C.c_int, [_P]),
    ("fluid_local_comm_create", C.c_int, None),
    ("fluid_rccl_unique_id", C.c_int, None),
    ("fluid_rccl_comm_create", C.c_int, None),
    ("fluid_rccl_comm_destroy", C.c_int, None),
    ("fluid_rccl_last_error", C.c_char_p, None),
]
for _name, _res, _args in SYMBOLS:
    _f = getattr(lib, _name)  # AttributeError here = header/library mismatch: fail loudly
    _f.restype = _res
    if _args is not None:
        _f.argtypes = _args


def check(rc):
    if rc != 0:
        raise FluidError(rc, lib.fluid_last_error().decode())
