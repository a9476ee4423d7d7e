#!/usr/bin/env python3
"""Idle time between consecutive kernels of one step, from a rocprofv3 --kernel-trace CSV (steps end at k_publish_dt):
python tools/step_gaps.py <kernel_trace.csv> [step index]   -> the largest gaps with the kernels either side, and a histogram"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_publish_dt' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 3
seg = rows[idx[which]:idx[which + 1] + 1]
name = lambda r: re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', '').replace('fl::', ''))[:44]
gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp']), name(a), name(b)) for a, b in zip(seg[:-1], seg[1:])]
tot = sum(g for g, _, _ in gaps)
print(f"{len(gaps)} gaps, {tot / 1e3:.1f} us idle in a step of {(int(seg[-1]['End_Timestamp']) - int(seg[0]['End_Timestamp'])) / 1e3:.1f} us")
for lo, hi in ((0, 1000), (1000, 2000), (2000, 5000), (5000, 20000), (20000, 10**9)):
    sel = [g for g, _, _ in gaps if lo <= g < hi]
    print(f"  gaps of {lo / 1e3:5.1f}..{min(hi, 99999000) / 1e3:7.1f} us: {len(sel):4d}, {sum(sel) / 1e3:8.1f} us")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:7.1f} us  {a}  ->  {b}")
