#!/bin/bash
# Fabric-side bytes (FETCH_SIZE x 2, WRITE_SIZE: MI355X guide, HBM section) of the HBM-proof stencil sweep, per kernel form:
# bash tools/pmc_stencil.sh   -> gpurun_out/pmc_stencil.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmst
mkdir -p $O
: > gpurun_out/pmc_stencil.txt
mkdir -p gpurun_out
for cfg in "fp64 20002 32" "fp64 0 0" "fp64 804 32" "fp32 20002 32" "fp32 0 0" "fp32 10402 16"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/run
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/run -- python3 tools/stencil_hbm.py $1 $2 $3 > $O/log.txt 2>&1 || { echo "failed $cfg $c"; tail -3 $O/log.txt; exit 1; }
    python3 - "$(ls -t $O/run/*/*_counter_collection.csv | head -n 1)" $c "$cfg" >> gpurun_out/pmc_stencil.txt <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and ("stencil" in r["Kernel_Name"] or "stream_probe" in r["Kernel_Name"]):
        rows[r["Kernel_Name"].split("(")[0].replace("void fl::", "")].append(float(r["Counter_Value"]))
for k, v in rows.items():
    scale = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0
    tail = v[len(v) // 2:]   # the timed launches (the first half warms every set up)
    print(f"{sys.argv[3]:18s} {k:40s} {sys.argv[2]:10s} {scale * 1024 * sum(tail) / len(tail) / 1e6:8.1f} MB per launch ({len(v)} launches)")
PY
  done
done
rm -rf $O
cat gpurun_out/pmc_stencil.txt
