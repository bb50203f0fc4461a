# Round-end validation on the GPU box: full GPU test-suite, the bench on every BASELINE config and on the fp32 arithmetic,
# rocprofv3 kernel statistics of the default bench command, smoke().  usage: gpurun -- bash benchmarks/final_validation.sh
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r02_final_pytest.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r02_final_pytest.txt
python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err || tail -5 gpurun_out/r02_final_bench.err
python bench.py --config causal --no-cpu-baseline > gpurun_out/r02_final_bench_causal.json 2>> gpurun_out/r02_final_bench.err
python bench.py --config c3 --no-cpu-baseline > gpurun_out/r02_final_bench_c3.json 2>> gpurun_out/r02_final_bench.err
python bench.py --arith fp32 --no-cpu-baseline > gpurun_out/r02_final_bench_fp32.json 2>> gpurun_out/r02_final_bench.err
python - <<'PY'
import json
for f in ("r02_final_bench", "r02_final_bench_causal", "r02_final_bench_c3", "r02_final_bench_fp32"):
    j = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, {k: j[k] for k in ("value", "ms_per_step", "host_issue_ms_per_step", "mean_loss")}, "dominant:", j["roofline"]["kernel"][:40], j["roofline"]["bound"], j["roofline"]["frac"], j["roofline"]["traffic"])
j = json.loads(open("gpurun_out/r02_final_bench.json").read().strip().splitlines()[-1])
print(j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
PY
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench_final -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-fp32-reference > $R/gpurun_out/prof_bench_final.log 2>&1
cd $R
head -8 gpurun_out/prof_bench_final/p_kernel_stats.csv | cut -c1-150
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
