cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "p2g" 2>&1 | tail -12
for pre in 2 195 445; do
  for f in crowd; do
    echo "== step $pre form $f"; FLUID_P2G_FORM=$f timeout -k 10 200 python tools/p2g_time.py 256 10 $pre || exit 1
  done
done
