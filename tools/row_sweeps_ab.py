#!/usr/bin/env python3
"""The 500-step 256^3 drop with the row-wise box sweeps on and off (FLUID_ROW_SWEEPS, kernels_grid.hip k_*4), twice each, interleaved:
mean ms/step per 50 steps and over the run.  Same arithmetic, so the runs are the same run (same iteration counts).
    python tools/row_sweeps_ab.py > profiles/rNN/row_sweeps_ab_256.txt"""
import os, re, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
for v in ("1", "0", "1", "0"):
    env = dict(os.environ, FLUID_ROW_SWEEPS=v)
    out = subprocess.run([sys.executable, os.path.join(HERE, "long_run.py"), "256", "500"], env=env, capture_output=True, text=True, timeout=300).stdout
    t, n, it = [], [], []
    for line in out.splitlines():
        m = re.match(r"steps\s+(\d+)-\s*(\d+):\s+([\d.]+) ms/step.*iters/solve\s+([\d.]+)", line)
        if m:
            t.append(float(m.group(3))); n.append(int(m.group(2)) - int(m.group(1)) + 1); it.append(float(m.group(4)))
    mean = sum(a * b for a, b in zip(t, n)) / max(1, sum(n))
    print(f"FLUID_ROW_SWEEPS={v}: mean {mean:.3f} ms/step   per 50 steps {t}   iters/solve {it}", flush=True)
