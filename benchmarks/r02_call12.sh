mkdir -p gpurun_out
python bench.py > gpurun_out/r02_c12_bench.json 2> gpurun_out/r02_c12_bench.err || tail -20 gpurun_out/r02_c12_bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02_c12_bench.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step','host_issue_ms_per_step','model_frac_of_f32_mfma_peak','mean_loss')})
r=j['roofline']; print(r['kernel'], r['achieved'], r['frac'], r['us_per_launch'])
for f in r['families']: print('  %-86s %5.1f/step %8.1f us %6.2f ms  %s %s' % (f['family'][:86], f['launches_per_step'], f['us_per_launch'], f['ms_per_step'], f.get('achieved'), f.get('frac')))
print(j['cpu_baseline'])
PY
for c in causal c3; do python bench.py --config $c --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r02_c12_bench_$c.json; python -c "
import json; j=json.loads(open('gpurun_out/r02_c12_bench_$c.json').read()); print('$c', j['value'], j['ms_per_step'], j['host_issue_ms_per_step'], j['roofline']['kernel'][:40], j['roofline']['frac'])"; done
