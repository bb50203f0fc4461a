# fresh-process bench runs alternating between the product library and an alternative build of the same ABI (CTN_LIB_PATH)
# usage: gpurun -- bash benchmarks/ab_libs.sh benchmarks/lab_h3_ar4.so [rounds] [bench args]
ALT=$1; N=${2:-3}; shift; shift
for i in $(seq $N); do
  for L in "" "$ALT"; do
    CTN_LIB_PATH=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-side-arith "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${L:-product}', j['value'], j['ms_per_step'], j['mean_loss'])"
  done
done
