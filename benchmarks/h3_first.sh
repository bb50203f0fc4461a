set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python benchmarks/h3_check.py > gpurun_out/h3_check.txt 2>&1 || { tail -30 gpurun_out/h3_check.txt; exit 1; }
cat gpurun_out/h3_check.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or composite" > gpurun_out/h3_t1.txt 2>&1 || true
tail -15 gpurun_out/h3_t1.txt
timeout -k 10 300 python bench.py --arith h3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/h3_bench.json 2> gpurun_out/h3_bench.err || { tail -20 gpurun_out/h3_bench.err; exit 1; }
cat gpurun_out/h3_bench.json
