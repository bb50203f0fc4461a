for x in 0 1; do
  CTN_GLN_FUSE=$x python bench.py --no-cpu-baseline --no-side-configs --no-side-arith 2>/dev/null > gpurun_out/bench_glnfuse_$x.json
  python - <<PY
import json
j=json.loads(open('gpurun_out/bench_glnfuse_$x.json').read().strip().splitlines()[-1])
print("gln_fuse=$x", j['value'], j['ms_per_step'])
for v in j['roofline']['families'][:11]:
    print("   ", v['family'][:70], v['launches_per_step'], v['us_per_launch'], v['ms_per_step'])
PY
done
