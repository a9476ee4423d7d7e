#!/bin/bash
# rocprofv3 kernel statistics of the snow-MPM step, the reference's scene and the scaled cone: bash tools/mpm_prof.sh -> gpurun_out/mpm_prof/
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/mpm_prof
mkdir -p $O
for cfg in "15 4 400 50 ref" "63 24 64 20 scaled"; do
  set -- $cfg
  python tools/mpm_run.py $1 $2 $3 $4 > $O/phases_$5.txt 2>&1
  rm -rf $O/t
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/mpm_run.py $1 $2 $3 $4 > $O/log_$5.txt 2>&1 || exit 1
  cp "$(ls -t $O/t/*/*_kernel_stats.csv | head -n 1)" $O/kernel_stats_$5.csv
  rm -rf $O/t
  cat $O/phases_$5.txt; head -12 $O/kernel_stats_$5.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://'
done
