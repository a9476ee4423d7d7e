#!/usr/bin/env python3
"""Soak run of the snow-MPM step on the reference's scene: 3000 steps, worst CG error, finiteness.  python tools/soak.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry


def main():
    fs = entry.load_package()
    sim = fs.MpmSim(); sim.upload_particles(fs.snow_cone())
    t0 = time.perf_counter(); worst = 0; mx = 0
    for i in range(3000):
        st = sim.step(); worst = max(worst, st["cg_error"]); mx = max(mx, st["cg_iters"])
    p = sim.particles(0); FE = sim.particles(2)
    print(f"mpm 3000 steps {time.perf_counter()-t0:.2f}s worst cg_error {worst:.2e} max iters {mx} finite {np.isfinite(p).all() and np.isfinite(FE).all()} |p|max {np.abs(p).max():.2f} num_active {st['num_active']}")
    sim.close()
    f = fs.FluidSim(n=128); f.upload_particles(fs.water_cube_drop(128, 8, seed=1))
    t0 = time.perf_counter(); its = 0
    for i in range(1500):
        st = f.step(); its += st["cg_iters"]
        assert st["relres"] < 2.3e-16 or st["num_active"] == 0, (i, st)
    p, v = f.download_particles()
    print(f"fluid 128^3 1500 steps {time.perf_counter()-t0:.2f}s iters {its} finite {np.isfinite(p).all() and np.isfinite(v).all()} inside {np.abs(p).max():.2f} unknowns {st['num_active']}")


if __name__ == "__main__":
    main()
