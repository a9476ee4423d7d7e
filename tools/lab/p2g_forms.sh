cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "p2g or forms_switch" 2>&1 | tail -3
for pre in 2 195 445; do
  echo "== step $pre default"; timeout -k 10 200 python tools/p2g_time.py 256 10 $pre || exit 1
done
