mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -q -m gpu -x > gpurun_out/r02_h_pytest.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r02_h_pytest.txt
ROUNDS=4 python benchmarks/ab_step.py "wgrad_chain=1" "wgrad_chain=0" 2>&1 | grep -v amdgpu.ids
CONFIG=causal ROUNDS=3 python benchmarks/ab_step.py "wgrad_chain=1" "wgrad_chain=0" 2>&1 | grep -v amdgpu.ids
