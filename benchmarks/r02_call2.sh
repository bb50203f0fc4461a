set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02_c2_pytest.txt 2>&1 || { tail -40 gpurun_out/r02_c2_pytest.txt; exit 1; }
tail -3 gpurun_out/r02_c2_pytest.txt
python benchmarks/gemm_lab.py pk > gpurun_out/r02_c2_lab.txt 2>&1; cat gpurun_out/r02_c2_lab.txt
CTN_PK_WGS=4 python benchmarks/gemm_lab.py pk_wgs4 > gpurun_out/r02_c2_lab4.txt 2>&1; cat gpurun_out/r02_c2_lab4.txt
CTN_PK_WGS=6 python benchmarks/gemm_lab.py pk_wgs6 > gpurun_out/r02_c2_lab6.txt 2>&1; cat gpurun_out/r02_c2_lab6.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_c2_bench.txt 2>&1; tail -1 gpurun_out/r02_c2_bench.txt | cut -c1-400
