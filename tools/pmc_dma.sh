#!/bin/bash
# Fabric-side bytes of the LDS-DMA stencil sweep (FETCH_SIZE x 2; WRITE_SIZE), per configuration: bash tools/pmc_dma.sh "cfg;cfg;..."
# cfg = "<prec> <variant> <cx>" as for tools/stencil_hbm.py  -> gpurun_out/pmc_dma.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmdma
mkdir -p $O
: > gpurun_out/pmc_dma.txt
IFS=';' read -ra CFGS <<< "${1:-fp64 936208 31000;fp64 936208 30000;fp64 936204 31000;fp64 936204 30000;fp64 0 0}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  for c in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
    rm -rf $O/run
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/run -- python3 tools/stencil_hbm.py $1 $2 $3 > $O/log.txt 2>&1 || { echo "failed $cfg $c"; tail -3 $O/log.txt; exit 1; }
    python3 - "$(ls -t $O/run/*/*_counter_collection.csv | head -n 1)" $c "$cfg" >> gpurun_out/pmc_dma.txt <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and ("stencil" in r["Kernel_Name"] or "stream_probe" in r["Kernel_Name"]):
        rows[r["Kernel_Name"].split("(")[0].replace("void fl::", "")].append(float(r["Counter_Value"]))
for k, v in rows.items():
    scale = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0
    tail = v[len(v) // 2:]   # the timed launches (the first half warms every set up)
    print(f"{sys.argv[3]:22s} {k:44s} {sys.argv[2]:10s} {scale * 1024 * sum(tail) / len(tail) / 1e6:8.1f} MB per launch ({len(v)} launches)")
PY
  done
done
rm -rf $O
cat gpurun_out/pmc_dma.txt
