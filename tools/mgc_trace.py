#!/usr/bin/env python3
"""Timeline of the last persistent coarse-level launch (k_mg_coarse) of a few bench-scene steps.
FLUID_MG_COARSE=1|2 python tools/mgc_trace.py [n] [steps]  -> per phase: tasks, first start, last end, mean wait / body / drain (us)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = os.path.join(ROOT, "gpurun_out", "mgc_trace.txt")
os.makedirs(os.path.dirname(path), exist_ok=True)
os.environ["FLUID_MGC_TRACE"] = path
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
for _ in range(steps):
    st = sim.step()
sim.close()
rows = [list(map(int, l.split())) for l in open(path)]
t0 = min(r[1] for r in rows)
# phases = runs of tasks separated by a wait: recover them from the stamps (a task of a new phase starts after every task of the one before ended)
print(f"tasks {len(rows)}  launch span {(max(r[4] for r in rows) - t0) / 100:.2f} us")
ph, cur = [], [rows[0]]
for r in rows[1:]:
    if r[2] >= max(q[4] for q in cur) - 5:   # its wait ended after everything before it had drained
        ph.append(cur); cur = [r]
    else:
        cur.append(r)
ph.append(cur)
for i, g in enumerate(ph):
    a = min(q[1] for q in g); w = min(q[2] for q in g); e = max(q[4] for q in g)
    print(f"phase {i}: tasks {len(g):4d}  first ticket {(a - t0) / 100:7.2f}  first start {(w - t0) / 100:7.2f}  last end {(e - t0) / 100:7.2f}  "
          f"mean body {sum(q[3] - q[2] for q in g) / len(g) / 100:6.2f}  max body {max(q[3] - q[2] for q in g) / 100:6.2f}  mean drain {sum(q[4] - q[3] for q in g) / len(g) / 100:5.2f}")
