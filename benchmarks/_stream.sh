python bench.py --no-cpu-baseline --no-side-arith --no-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['config'], {k:(v['value'],v['ms_per_step']) for k,v in j['other_configs'].items()}, j['other_configs']['causal'].get('streaming_inference'))"
ROUNDS=3 STEPS=10 python benchmarks/ab_step.py "side=1" 2>&1 | grep median
