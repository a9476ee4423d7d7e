cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fin
timeout -k 10 600 python bench.py > gpurun_out/fin/bench.json 2> gpurun_out/fin/bench.err
timeout -k 10 300 python bench.py --force-dist --no-cpu --no-micro --no-long-run --no-mpm > gpurun_out/fin/bench_force_dist.json 2> /dev/null
timeout -k 10 300 python bench.py --no-cpu --no-micro --no-long-run --no-mpm > gpurun_out/fin/bench_single_same_run.json 2> /dev/null
