mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "wgrad or temporal_block or model_matches" > gpurun_out/r02_b3_pytest.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r02_b3_pytest.txt
python benchmarks/b3_only.py W1; python benchmarks/b3_only.py W2
ROUNDS=3 python benchmarks/ab_step.py "b3_wgrad_blocks=256" "b3_wgrad_blocks=512" 2>&1 | grep -v amdgpu.ids
