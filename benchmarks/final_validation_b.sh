T=${1:-r04_final2}
timeout -k 10 200 python -m pytest tests/test_gpu_h3.py -q -m gpu -k trajectories -s 2>&1 | grep -E "passed|failed|10 steps" | cut -c1-1200
python bench.py --config causal --no-cpu-baseline > gpurun_out/${T}_bench_causal.json 2>> gpurun_out/${T}_bench.err
python bench.py --config c3 --no-cpu-baseline > gpurun_out/${T}_bench_c3.json 2>> gpurun_out/${T}_bench.err
python bench.py --arith fp32 --no-cpu-baseline --no-side-arith --no-side-configs > gpurun_out/${T}_bench_fp32.json 2>> gpurun_out/${T}_bench.err
python bench.py --arith b6 --no-cpu-baseline --no-side-arith --no-side-configs > gpurun_out/${T}_bench_b6.json 2>> gpurun_out/${T}_bench.err
T=$T python - <<EOF
import json, os
T = os.environ["T"]
for f in ("bench_causal", "bench_c3", "bench_fp32", "bench_b6"):
    j = json.loads(open("gpurun_out/%s_%s.json" % (T, f)).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f, {k: j[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step", "mean_loss", "gemm_arith")}, "dominant:", r["kernel"][:40], r["bound"], r["frac"], "step:", r["step"]["hbm_frac"], r["step"]["hbm_frac_of_fused_minimum"], r["step"]["bf16_mfma_frac"])
EOF
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T} -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-side-arith --no-side-configs > $R/gpurun_out/prof_${T}.log 2>&1
cd $R
cp $(ls gpurun_out/prof_${T}/*kernel_stats.csv gpurun_out/prof_${T}/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats_bench_steps5.csv
python benchmarks/kstats.py gpurun_out/prof_${T} 7 16
bash benchmarks/pmc_traffic.sh pw_wgrad_b3_kernel benchmarks/gemm_only.py gpurun_out/r04_pmc_h3_wgrad_dW2_pro.json "B2 weight gradient dW2 (gLN prologue)" "W2 30" h3 | tail -12
bash benchmarks/pmc_traffic.sh pw_gemm_b3p_kernel benchmarks/gemm_only.py gpurun_out/r04_pmc_h3_B1.json "B1 input gradient W2^T.dout (+ gLN backward sums)" "B1 30" h3 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
