cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/p2gprof
FLUID_P2G_FORM=crowd timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p2gprof -- python3 tools/p2g_time.py 256 10 ${STEP:-445} > gpurun_out/p2gprof.log 2>&1
python3 - "$(ls -t gpurun_out/p2gprof/*/*_kernel_trace.csv | head -n 1)" <<'PY' > gpurun_out/p2gprof.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 10 launches of each p2g kernel
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'p2g' in n: d[n.split('(')[0].replace('void ','').replace('fl::','')].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, v in d.items():
    print(f"{k:40s} n={len(v):5d} last10 avg {sum(v[-10:])/10:9.1f} us")
PY
rm -rf gpurun_out/p2gprof
