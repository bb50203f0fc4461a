mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "dw_plain or temporal_block" > gpurun_out/r02_pytest_dw.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r02_pytest_dw.txt
for d in 1 64 128; do python benchmarks/dw_bwd_only.py $d 2>&1 | grep -v amdgpu.ids; done
ROUNDS=3 python benchmarks/ab_step.py "arith=1" 2>&1 | grep -v amdgpu.ids
