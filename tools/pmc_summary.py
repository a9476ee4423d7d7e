#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM bytes per kernel.

usage: tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv>
Prints a JSON object {kernel: {fetch_bytes, write_bytes, bytes_per_launch, launches}}.
Counters are in KB; FETCH_SIZE is doubled (gfx950 correction, see profiles/r01/pmc_traffic.json
"method").  Multigrid kernels run on several levels under one name: only the launches with the largest
grid are kept ("(largest level)": level 0 for k_mg_up / k_mg_down<..., false>, level 1 for k_mg_down<..., true>).
"""
import csv, json, sys, collections

def load(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        rows[r["Kernel_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return rows

def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("fl::", "")
    return name.split("(")[0]

def fold(rows, scale):
    out = {}
    for k, v in rows.items():
        s = short(k)
        top = max(g for g, _ in v)
        if "k_mg_" in s and "tail" not in s and len({g for g, _ in v}) > 1:
            v = [x for x in v if x[0] == top]
            s += " (largest level)"
        out[s] = (scale * 1024.0 * sum(c for _, c in v) / len(v), len(v))
    return out

def main():
    f = fold(load(sys.argv[1], "FETCH_SIZE"), 2.0)
    w = fold(load(sys.argv[2], "WRITE_SIZE"), 1.0)
    res = {}
    for k in sorted(f):
        if k not in w or k.startswith("__amd") or "at::" in k:
            continue
        res[k] = {"fetch_bytes": f[k][0], "write_bytes": w[k][0],
                  "bytes_per_launch": f[k][0] + w[k][0], "launches": f[k][1]}
    json.dump(res, sys.stdout, indent=1)

if __name__ == "__main__":
    main()
