#!/usr/bin/env python3
"""Time the particle -> grid phase alone on the bench scene: python tools/p2g_time.py [n] [reps] [steps before]
(two full steps first, then fluid_p2g() repeatedly on the sorted particles; hipEvent time per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
fs = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sim = fs.FluidSim(n=n)
sim.upload_particles(fs.water_cube_drop(n, 8, seed=0))
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for _ in range(pre):
    sim.step()
sim.p2g()
sim.profile_reset()
sim.profile_enable(1)
for _ in range(reps):
    sim.p2g()
sim.profile_enable(0)
p = sim.profile_read(fs.PROF.P2G)
sim.profile_reset()
sim.profile_enable(1)
for _ in range(5):
    sim.step()
sim.profile_enable(0)
q = sim.profile_read(fs.PROF.SORT)
print(f"p2g {p['total_ms'] / max(p['sampled'], 1) * 1e3:.1f} us over {p['sampled']} calls; sort {q['total_ms'] / max(q['sampled'], 1) * 1e3:.1f} us")
