set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k composite > gpurun_out/r02_c1_composite.txt 2>&1 || { tail -30 gpurun_out/r02_c1_composite.txt; exit 1; }
tail -3 gpurun_out/r02_c1_composite.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_c1_bench.txt 2>&1; tail -2 gpurun_out/r02_c1_bench.txt
CTN_COMPOSITE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_c1_bench_nocomp.txt 2>&1; tail -2 gpurun_out/r02_c1_bench_nocomp.txt
./benchmarks/mfma_probe.bin > gpurun_out/r02_c1_probe.txt 2>&1; cat gpurun_out/r02_c1_probe.txt
python benchmarks/gemm_lab.py base > gpurun_out/r02_c1_lab.txt 2>&1
CTN_LIB_PATH=$PWD/benchmarks/lab/libctn_SKIP_MAIN.so python benchmarks/gemm_lab.py skip_main >> gpurun_out/r02_c1_lab.txt 2>&1
CTN_LIB_PATH=$PWD/benchmarks/lab/libctn_SKIP_EPI.so python benchmarks/gemm_lab.py skip_epi >> gpurun_out/r02_c1_lab.txt 2>&1
cat gpurun_out/r02_c1_lab.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r02_c1_pytest.txt 2>&1; tail -5 gpurun_out/r02_c1_pytest.txt
