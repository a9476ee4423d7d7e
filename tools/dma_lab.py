#!/usr/bin/env python3
"""Developer sweep of the LDS-DMA plane-ring stencil (kernels_stencil.hip): python tools/dma_lab.py [check] [n]
Checks the form bit for bit against the tiled kernel at a few sizes, then times it from HBM (launches rotating over > 1 GiB of
separate operand sets) next to the register-staged default and the copy probe.
variant = 900000 + D*10000 + RY ; cx = planes per chunk (0: N/8 + 1)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
F = fs.FIELD


def setup(n, prec, holes):
    sim = fs.FluidSim(n=n, precision=prec)
    solid = sim.field(F.SOLID)
    rng = np.random.default_rng(n)
    cont = ((solid == 0) & ((rng.random((n, n, n)) < 0.8) if holes else True)).astype(np.float32)
    sim.upload_field(F.CONTAINER, cont)
    sim.flags_index()
    s = rng.uniform(-1, 1, size=(n, n, n)).astype(np.float64 if prec == "fp64" else np.float32)
    if not holes:
        s *= (solid == 0)
    sim.upload_field(F.SEARCH, s)
    return sim


def run(sim, var, cx, hbm=False, reps=28):
    os.environ["FLUID_MARCH_VARIANT"] = str(var)
    os.environ["FLUID_MARCH_CX"] = str(cx)
    if hbm:
        return min(sim.stencil_apply_hbm(reps=reps, box=0, footprint_bytes=1 << 30)[0] for _ in range(2))
    return sim.stencil_apply(reps=1, box=0)


def check():
    bad = 0
    for n in (64, 96, 48, 112, 256):
        for prec in ("fp64", "fp32"):
            sim = setup(n, prec, True)
            os.environ.pop("FLUID_MARCH_VARIANT", None)
            sim.stencil_apply(reps=1, box=2)
            want = sim.field(F.Q)
            for var, cx in ((900000, 0), (920004, 7), (940008, 33), (930016, 5), (960005, 31), (930008, 32), (40404, 9)):
                sim.upload_field(F.Q, np.zeros_like(want)) if False else None
                run(sim, var, cx)
                got = sim.field(F.Q)
                ok = np.array_equal(got, want)
                bad += not ok
                print(n, prec, var, cx, "ok" if ok else f"MISMATCH {np.count_nonzero(got != want)} cells", flush=True)
            sim.close()
    return bad


def sweep(n):
    for prec, T in (("fp64", 8), ("fp32", 4)):
        sim = setup(n, prec, False)
        algo = n ** 3 * (2 * T + 1)
        rows = [(0, 0), (20002, 4), (40404, 64 if prec == "fp64" else 32)]
        if len(sys.argv) > 2:
            rows = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
        else:
            for d in (2, 3, 4):
                for cx in (32, 33, 34):
                    rows.append((900008 + d * 10000, cx))
            rows += [(930004, 33), (930016, 33)]
        for var, cx in rows:
            ms = run(sim, var, cx, hbm=True)
            print(prec, var, cx, f"{ms*1e3:.1f} us {algo/ms/1e6:.0f} GB/s  {algo/ms/1e6/8000:.3f}", flush=True)
        sim.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "check":
        sys.exit(1 if check() else 0)
    sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
