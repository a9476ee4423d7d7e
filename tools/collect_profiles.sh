#!/bin/bash
# Everything under profiles/ for one round, on the GPU box: bash tools/collect_profiles.sh  (writes gpurun_out/fin/)
# rocprofv3: kernel trace and the two PMC passes are separate runs; the program follows `--` directly.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/fin
mkdir -p $O
PART=${1:-all}
if [ $PART = all ] || [ $PART = 1 ]; then
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu --no-long-run --no-mpm > $O/trace.log 2>&1
python tools/step_breakdown.py "$(ls -t $O/trace/*/*_kernel_trace.csv | head -n 1)" 12 > $O/step_breakdown.txt
cp "$(ls -t $O/trace/*/*_kernel_stats.csv | head -n 1)" $O/kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-micro --no-long-run --no-mpm > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-micro --no-long-run --no-mpm > $O/write.log 2>&1
python tools/pmc_summary.py "$(ls -t $O/fetch/*/*_counter_collection.csv | head -n 1)" "$(ls -t $O/write/*/*_counter_collection.csv | head -n 1)" > $O/pmc.json
timeout -k 10 300 python bench.py --force-dist --no-cpu --no-micro --no-long-run --no-mpm > $O/bench_force_dist.json 2> /dev/null
fi
if [ $PART = all ] || [ $PART = 2 ]; then
timeout -k 10 300 python bench.py --n 128 --no-long-run --no-mpm > $O/bench128.json 2> /dev/null
timeout -k 10 300 python bench.py --n 512 --ppc 4 --steps 5 --warmup 2 --no-cpu --no-micro --no-long-run --no-mpm > $O/bench512.json 2> /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/splash -- python3 tools/long_run.py 256 200 > $O/splash.log 2>&1
python tools/step_breakdown.py "$(ls -t $O/splash/*/*_kernel_trace.csv | head -n 1)" 195 > $O/step_breakdown_splash.txt
rm -f $O/splash/*/*_kernel_trace.csv
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/settled -- python3 tools/long_run.py 256 450 > $O/settled.log 2>&1
python tools/step_breakdown.py "$(ls -t $O/settled/*/*_kernel_trace.csv | head -n 1)" 445 > $O/step_breakdown_settled.txt
timeout -k 10 300 python tools/long_run.py 256 500 > $O/long_run.txt 2>&1
fi
if [ $PART = all ] || [ $PART = 3 ]; then
# the snow-MPM step: kernel statistics of both scenes, and the fabric-side bytes of the operator's kernels on the scaled cone
bash tools/mpm_prof.sh > $O/mpm_prof.log 2>&1
cp gpurun_out/mpm_prof/kernel_stats_ref.csv $O/mpm_kernel_stats_ref_scene.csv; cp gpurun_out/mpm_prof/kernel_stats_scaled.csv $O/mpm_kernel_stats_scaled.csv
cat gpurun_out/mpm_prof/phases_ref.txt gpurun_out/mpm_prof/phases_scaled.txt > $O/mpm_phases.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/mpmc
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/mpmc -- python3 tools/mpm_run.py 63 24 64 6 > $O/mpmc_$c.log 2>&1
  python3 - "$(ls -t $O/mpmc/*/*_counter_collection.csv | head -n 1)" $c >> $O/mpm_pmc.txt <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and "k_mpm_" in r["Kernel_Name"]:
        rows[r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]].append(float(r["Counter_Value"]))
scale = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:28s} {sys.argv[2]:10s} {scale * 1024 * sum(v) / len(v) / 1e6:9.3f} MB per launch ({len(v)} launches; FETCH_SIZE doubled: gfx950 correction)")
PY
  rm -rf $O/mpmc
done
bash tools/pmc_dma.sh "fp64 0 0;fp32 0 0;fp64 40404 64;fp32 40404 32;fp64 20002 4;fp32 20002 4" > $O/pmc_stencil.log 2>&1; cp gpurun_out/pmc_dma.txt $O/pmc_stencil_dma.txt
# the droplets: on / off from the same late states, and every claimed set against scipy's labelling
timeout -k 10 300 python tools/droplet_ab.py 256 200 300 400 445 490 > $O/droplet_ab.txt 2>&1
timeout -k 10 300 python tools/droplet_check.py 256 460 115 > $O/droplet_check.txt 2>&1
timeout -k 10 300 python tools/spray_stats.py 256 445 > $O/spray_stats_445.txt 2>&1
fi
# the raw traces are large: only the summaries travel back
rm -f $O/trace/*/*_kernel_trace.csv $O/splash/*/*_kernel_trace.csv $O/settled/*/*_kernel_trace.csv $O/fetch/*/*_counter_collection.csv $O/write/*/*_counter_collection.csv
ls -la $O
