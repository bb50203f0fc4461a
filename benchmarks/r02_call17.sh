mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -x -q -m gpu -k "cln or cLN or causal or model or temporal_block or assorted or streaming or solver" > gpurun_out/r02_c17_pytest.txt 2>&1 || { tail -40 gpurun_out/r02_c17_pytest.txt; exit 1; }
tail -2 gpurun_out/r02_c17_pytest.txt
CONFIG=causal ROUNDS=3 python benchmarks/ab_step.py "cln_side=1" "cln_side=0" 2>&1 | grep -v amdgpu.ids
python bench.py --config causal --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r02_c17_bench_causal.json
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02_c17_bench_causal.json').read())
print(j['value'], j['ms_per_step'], j['host_issue_ms_per_step'])
for f in j['roofline']['families'][:10]: print('  %-80s %5.1f/step %8.1f us %6.2f ms' % (f['family'][:80], f['launches_per_step'], f['us_per_launch'], f['ms_per_step']))
PY
