"""MI355X-native PIC/FLIP step: Python host-side mirror of the reference's step surface.

The reference (Aakash1312/Fluid-Simulation, ``fluid.cc``) is one ``main()`` whose loop body
(``fluid.cc:1368-1507``) is the hot path.  This package binds the C ABI of
``libfluid_hip.so`` (``include/fluid_hip.h``) with ctypes: plumbing only — all compute is
hand-written HIP for gfx950.  There is NO CPU fallback: if the shared library is missing,
import fails; if no GPU is visible, ``FluidSim(...)`` raises.
"""
from ._lib import lib, FluidError, Params, StepStats, FIELD, PROF, MpmParams, MpmStepStats  # noqa: F401
from .mpm import MpmSim, snow_cone, mpm_eval, MPM_P, MPM_F  # noqa: F401
from .sim import FluidSim, water_cube_drop, reference_scatter, grid_bounds, write_vdb, VdbStream  # noqa: F401

def load_dist():
    """torch is imported only when the multi-GPU path is used."""
    from . import dist
    return dist


__all__ = ["MpmParams", "MpmStepStats", "MpmSim", "snow_cone", "mpm_eval", "MPM_P", "MPM_F", "load_dist", "FluidSim", "FluidError", "Params", "StepStats", "FIELD", "PROF", "water_cube_drop", "reference_scatter", "grid_bounds", "write_vdb", "VdbStream", "lib"]
