set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pm
mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-long-run > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-long-run > $O/write.log 2>&1
python tools/pmc_summary.py "$(ls -t $O/fetch/*/*_counter_collection.csv | head -n 1)" "$(ls -t $O/write/*/*_counter_collection.csv | head -n 1)" > $O/pmc_micro.json
rm -f $O/fetch/*/*_counter_collection.csv $O/write/*/*_counter_collection.csv
python -c "
import json; d=json.load(open('$O/pmc_micro.json')); print({k: round(v['bytes_per_launch']/1e6,1) for k,v in d.items() if 'stencil' in k})"
