# Round-end validation on the GPU box: full GPU test-suite, the bench on every BASELINE config and on the other arithmetics,
# rocprofv3 kernel statistics of the default bench command, PMC traffic of the dominant kernel, smoke().
# usage: gpurun -- bash benchmarks/final_validation.sh [tag]      (files: gpurun_out/<tag>_*; default tag r04_final)
T=${1:-r04_final}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/${T}_pytest.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/${T}_pytest.txt
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || tail -5 gpurun_out/${T}_bench.err
python bench.py --config causal --no-cpu-baseline > gpurun_out/${T}_bench_causal.json 2>> gpurun_out/${T}_bench.err
python bench.py --config c3 --no-cpu-baseline > gpurun_out/${T}_bench_c3.json 2>> gpurun_out/${T}_bench.err
python bench.py --arith fp32 --no-cpu-baseline --no-side-arith --no-side-configs > gpurun_out/${T}_bench_fp32.json 2>> gpurun_out/${T}_bench.err
python bench.py --arith b6 --no-cpu-baseline --no-side-arith --no-side-configs > gpurun_out/${T}_bench_b6.json 2>> gpurun_out/${T}_bench.err
T=$T python - <<'PY'
import json, os
T = os.environ["T"]
for f in ("bench", "bench_causal", "bench_c3", "bench_fp32", "bench_b6"):
    j = json.loads(open("gpurun_out/%s_%s.json" % (T, f)).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f, {k: j[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step", "mean_loss", "gemm_arith", "dtype")}, "dominant:", r["kernel"][:40], r["bound"], r["frac"], r["traffic"],
          "step:", r["step"]["hbm_frac"], r["step"]["bf16_mfma_frac"])
j = json.loads(open("gpurun_out/%s_bench.json" % T).read().strip().splitlines()[-1])
print({k: j["cpu_baseline"][k] for k in ("value", "cores", "warmup_steps", "timed_steps", "median_s_per_step", "min_s_per_step")}, j["cpu_baseline"].get("threads8"))
print("inference", {k: j["inference"][k] for k in ("value", "ms_per_forward")}, j["inference"]["roofline"]["kernel"][:40], j["inference"]["roofline"]["frac"], j["inference"]["roofline"]["step"])
for k, v in j["other_configs"].items():
    print(k, {q: v[q] for q in ("value", "ms_per_step", "hbm_frac_of_fused_minimum")}, v.get("streaming_inference"))
PY
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T} -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-side-arith --no-side-configs > $R/gpurun_out/prof_${T}.log 2>&1
cd $R
cp $(ls gpurun_out/prof_${T}/*kernel_stats.csv gpurun_out/prof_${T}/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats_bench_steps5.csv
python benchmarks/kstats.py gpurun_out/prof_${T} 7 14
bash benchmarks/pmc_traffic.sh pw_wgrad_b3_kernel benchmarks/gemm_only.py gpurun_out/${T%%_*}_pmc_h3_wgrad_dW2_pro.json "B2 weight gradient dW2 (gLN prologue)" "W2 30" h3 | tail -12
bash benchmarks/pmc_traffic.sh pw_wgrad_b3_kernel benchmarks/gemm_only.py gpurun_out/${T%%_*}_pmc_h3_wgrad_dW1.json "B6 weight gradient dW1 + slab_reduce" "W1 30" h3 | tail -4
bash benchmarks/pmc_traffic.sh pw_gemm_b3p_kernel benchmarks/gemm_only.py gpurun_out/${T%%_*}_pmc_h3_K3.json "K3 1x1 H->B (gLN prologue + residual)" "K3 30" h3 | tail -4
bash benchmarks/pmc_traffic.sh pw_gemm_b3p_kernel benchmarks/gemm_only.py gpurun_out/${T%%_*}_pmc_h3_K1.json "K1 1x1 B->H (+ PReLU/gLN statistics)" "K1 30" h3 | tail -4
bash benchmarks/pmc_traffic.sh pw_gemm_b3p_kernel benchmarks/gemm_only.py gpurun_out/${T%%_*}_pmc_h3_B1.json "B1 input gradient W2^T.dout (+ gLN backward sums)" "B1 30" h3 | tail -4
bash benchmarks/pmc_traffic.sh dw_bwd_kernel benchmarks/dw_bwd_only.py gpurun_out/${T%%_*}_pmc_dw_bwd.json "B3 dw_bwd fused (gLN2'.PReLU2'.dw^T)" "8" any | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
