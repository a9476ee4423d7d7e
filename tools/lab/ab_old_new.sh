set -e
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in new old; do
  if [ $v = old ]; then cp fluid-simulation_amd/libfluid_hip.so /tmp/new.so; cp fluid-simulation_amd/libfluid_hip_old.so fluid-simulation_amd/libfluid_hip.so; fi
  echo "== $v 195"; timeout -k 10 200 python tools/settled_time.py 256 195
  echo "== $v 445"; timeout -k 10 200 python tools/settled_time.py 256 445
  if [ $v = old ]; then cp /tmp/new.so fluid-simulation_amd/libfluid_hip.so; fi
done
done
