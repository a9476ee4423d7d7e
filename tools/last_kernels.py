#!/usr/bin/env python3
"""Durations of the last launches of the kernels whose name contains a pattern, from a rocprofv3 --kernel-trace CSV:
python tools/last_kernels.py <kernel_trace.csv> <pattern> [count]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-(int(sys.argv[3]) if len(sys.argv) > 3 else 12):]:
    print(f'{r["Kernel_Name"].split("(")[0][-60:]:60s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:10.1f} us  grid {r.get("Grid_Size", r.get("Grid_Size_X", "?"))}')
