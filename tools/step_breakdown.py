#!/usr/bin/env python3
"""Per-kernel time of one steady-state step from a rocprofv3 --kernel-trace CSV (steps end at k_publish_dt)."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_publish_dt' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 3
a, b = idx[which], idx[which + 1]
seg = rows[a + 1:b + 1]
d = collections.OrderedDict()
for r in seg:
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', '').replace('fl::', ''))
    t = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    c = d.setdefault(n, [0, 0]); c[0] += 1; c[1] += t
tot = sum(v[1] for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} {v[0]:4d} {v[1]/1e3:9.1f} us  avg {v[1]/v[0]/1e3:7.2f}  {100*v[1]/tot:5.1f}%")
print(f"kernels {len(seg)}  busy {tot/1e3:.1f} us  span {(int(seg[-1]['End_Timestamp'])-int(rows[a]['End_Timestamp']))/1e3:.1f} us")
