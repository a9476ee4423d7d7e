#!/bin/bash
# A/B of the four-groups-per-tile coarse legs on the GPU box: bash tools/ab_ng4.sh
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/ng4
mkdir -p $O
for v in 0 1; do
  FLUID_MG_NG4=$v timeout -k 10 200 python bench.py --no-cpu --no-micro --no-long-run --no-mpm > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("ng4=$v", round(d["value"],1), round(d["ms_per_step"],3), d["step_stats"]["cg_iters_total"])
PY
done
FLUID_MG_NG4=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --no-cpu --no-micro --no-long-run --no-mpm > $O/trace.log 2>&1 || exit 1
python tools/step_breakdown.py "$(ls -t $O/trace/*/*_kernel_trace.csv | head -n 1)" 12 > $O/step_breakdown.txt
rm -rf $O/trace
head -14 $O/step_breakdown.txt; tail -1 $O/step_breakdown.txt
