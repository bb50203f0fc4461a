mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r02_e_pytest.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r02_e_pytest.txt
python bench.py > gpurun_out/r02_e_bench.json 2> gpurun_out/r02_e_bench.err || tail -5 gpurun_out/r02_e_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r02_e_bench.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step", "mean_loss")})
print("dominant:", j["roofline"]["kernel"], j["roofline"]["bound"], j["roofline"]["frac"], j["roofline"]["traffic"])
for r in j["roofline"]["families"][:13]:
    print("%-70s %5.1f/step %7.2f us %6.3f ms  %s %s frac %s" % (r["family"][:70], r["launches_per_step"], r["us_per_launch"], r["ms_per_step"], r.get("bound"), r.get("achieved"), r.get("frac")))
PY
bash benchmarks/pmc_traffic.sh 'pw_wgrad_b3_kernel<1' benchmarks/b3_only.py gpurun_out/r02_pmc_b3_wgrad_dW2_pro.json "B2 pw_wgrad<PRO> + slab_reduce (dW2 = dout . gLN2(prelu(d))^T)" W2 b3 > /dev/null
bash benchmarks/pmc_traffic.sh 'pw_wgrad_b3_kernel<0' benchmarks/b3_only.py gpurun_out/r02_pmc_b3_wgrad_dW1.json "B6 pw_wgrad + slab_reduce (dW1 = dh1 . x^T)" W1 b3 > /dev/null
grep hbm_bytes gpurun_out/r02_pmc_b3_wgrad_*.json
CONFIG=causal ROUNDS=2 python benchmarks/ab_step.py "arith=1" "arith=0" 2>&1 | grep -v amdgpu.ids
CONFIG=c3 ROUNDS=2 python benchmarks/ab_step.py "arith=1" "arith=0" 2>&1 | grep -v amdgpu.ids
