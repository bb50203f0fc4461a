mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r02_b3_pytest.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r02_b3_pytest.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_b3_bench.json 2> gpurun_out/r02_b3_bench.err || tail -5 gpurun_out/r02_b3_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r02_b3_bench.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step", "dtype", "mean_loss")})
for r in j["roofline"]["families"][:12]:
    print("%-78s %5.1f/step %7.2f us %6.3f ms  %s %s frac %s" % (r["family"][:78], r["launches_per_step"], r["us_per_launch"], r["ms_per_step"], r.get("bound"), r.get("achieved"), r.get("frac")))
print(j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
PY
