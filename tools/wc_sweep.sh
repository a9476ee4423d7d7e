# PCG iterations against the weights of the coarse corrections (FLUID_MG_WC = level 0, level 1, deeper kernel levels, tail)
for wc in "1.2,1.1,1,1" "1.25,1.1,1,1" "1.2,1.15,1,1" "1.2,1.1,1.05,1" "1.3,1.1,1,1" "1.2,1.2,1,1"; do
  a=$(FLUID_MG_WC=$wc timeout -k 10 200 python bench.py --no-cpu --no-micro --steps 10 --n 128 | python -c "import json,sys; d=json.load(sys.stdin); print(d['step_stats']['cg_iters_total'])")
  b=$(FLUID_MG_WC=$wc timeout -k 10 200 python bench.py --no-cpu --no-micro --steps 10 | python -c "import json,sys; d=json.load(sys.stdin); print(d['step_stats']['cg_iters_total'])")
  c=$(FLUID_MG_WC=$wc timeout -k 10 300 python bench.py --n 512 --ppc 4 --steps 5 --warmup 2 --no-cpu --no-micro | python -c "import json,sys; d=json.load(sys.stdin); print(d['step_stats']['cg_iters_total'])")
  echo "$wc  128: $a  256: $b  512: $c"
done
