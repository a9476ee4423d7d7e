#!/usr/bin/env python3
"""How well do two latency-bound solves share the GPU?  Two handles stepped from two host threads (own streams) against one
alone: python tools/concurrency_probe.py [n] [steps] [skip]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 160


def make():
    s = fs.FluidSim(n=n)
    s.upload_particles(fs.water_cube_drop(n, 8, seed=0))
    for _ in range(skip):
        s.step()
    return s


def run(s, out, k):
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step()
    out[k] = time.perf_counter() - t0


a, b = make(), make()
out = [0, 0]
run(a, out, 0)
alone = out[0]
b2 = make()   # same state as `a` had before its timed steps
a.close()
a = make()
ta, tb = threading.Thread(target=run, args=(a, out, 0)), threading.Thread(target=run, args=(b2, out, 1))
t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); both = time.perf_counter() - t0
print(f"n={n} steps {skip}..{skip + steps}: one handle {alone / steps * 1e3:.2f} ms/step; two concurrent handles {both / steps * 1e3:.2f} ms per pair of steps "
      f"({both / alone:.2f} x one)")
