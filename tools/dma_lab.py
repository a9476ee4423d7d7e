#!/usr/bin/env python3
"""Developer sweep of the LDS-DMA plane-ring stencil (kernels_stencil.hip): python tools/dma_lab.py [check] [n]
Checks the form bit for bit against the tiled kernel at a few sizes, then times it from HBM (launches rotating over > 1 GiB of
separate operand sets) next to the register-staged default and the copy probe.
variant = 900000 + D*10000 + G*1000 + NP*100 + RY ; cx = cxlen + 1000*nt + 10000*rot + 100000*mode"""
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
            for var, cx in ((936208, 31000), (926204, 11000), (933208, 1000 + 7), (936408, 50000), (934208, 21000 + 5), (946208, 91000), (936216, 31000), (936208, 31003)):
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
        rows = [(0, 0), (20002, 4)]
        if len(sys.argv) > 2:
            rows = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
        else:
            for rot in (0, 1, 3):
                for ry in (4, 8):
                    rows.append((936200 + ry, 1000 + rot * 10000))
        for var, cx in rows:
            ms = run(sim, var, cx, hbm=True)
            print(prec, var, cx, f"{ms*1e3:.1f} us {algo/ms/1e6:.0f} GB/s  {algo/ms/1e6/8000:.3f}", flush=True)
        sim.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "check":
        sys.exit(1 if check() else 0)
    sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
