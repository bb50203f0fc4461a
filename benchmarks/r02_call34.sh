mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -q -m gpu -x > gpurun_out/r02_i_pytest.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r02_i_pytest.txt; grep -E "^E  " gpurun_out/r02_i_pytest.txt | head -8
ROUNDS=4 python benchmarks/ab_step.py "fuse_b4=1" "fuse_b4=0" 2>&1 | grep -v amdgpu.ids
