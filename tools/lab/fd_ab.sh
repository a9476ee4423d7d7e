cd $GRAFT_REPO_ROOT
for r in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu --no-micro --no-long-run --no-mpm 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single', d['value'], d['ms_per_step'])"
  timeout -k 10 200 python bench.py --force-dist --no-cpu --no-micro --no-long-run --no-mpm 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dist  ', d['value'], d['ms_per_step'])"
done
exit 0
rm -rf gpurun_out/fdtrace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fdtrace -- python3 bench.py --force-dist --no-cpu --no-micro --no-long-run --no-mpm > gpurun_out/fdtrace.log 2>&1
python tools/step_breakdown.py "$(ls -t gpurun_out/fdtrace/*/*_kernel_trace.csv | head -n 1)" 12 > gpurun_out/step_breakdown_fd.txt
rm -rf gpurun_out/fdtrace
