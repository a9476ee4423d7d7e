#!/bin/bash
# A/B of the persistent coarse-level launch on the GPU box: bash tools/ab_coarse.sh  (writes gpurun_out/ab/)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/ab
mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu --no-micro --no-long-run --no-mpm > $O/bench_$name.json 2> $O/bench_$name.err || return 1
  python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
print("$name", round(d["value"],1), round(d["ms_per_step"],3), d.get("step_stats",{}).get("cg_iters_total"))
PY
}
run mode0 FLUID_MG_COARSE=0 || exit 1
run mode1 FLUID_MG_COARSE=1 || exit 1
run mode1_b128 FLUID_MG_COARSE=1 FLUID_MG_COARSE_BLOCKS=128 || exit 1
run mode2 FLUID_MG_COARSE=2 || exit 1
for m in 0 1; do
  FLUID_MG_COARSE=$m timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$m -- python3 bench.py --no-cpu --no-micro --no-long-run --no-mpm > $O/trace$m.log 2>&1 || exit 1
  python tools/step_breakdown.py "$(ls -t $O/trace$m/*/*_kernel_trace.csv | head -n 1)" 12 > $O/step_breakdown_mode$m.txt
  python tools/step_gaps.py "$(ls -t $O/trace$m/*/*_kernel_trace.csv | head -n 1)" 12 > $O/step_gaps_mode$m.txt 2>&1
  rm -rf $O/trace$m
  head -12 $O/step_breakdown_mode$m.txt; tail -1 $O/step_breakdown_mode$m.txt
done
