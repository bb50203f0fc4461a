mkdir -p gpurun_out
python benchmarks/b3_check.py 2>&1 | grep -v amdgpu.ids | grep "b3_tile" | tee gpurun_out/r02_b3_check2.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -q -m gpu -x > gpurun_out/r02_b3_pytest.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r02_b3_pytest.txt
ROUNDS=3 python benchmarks/ab_step.py "arith=1,b3_tile=1" "arith=1,b3_tile=2" "arith=1,b3_tile=0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_b3_ab.txt
