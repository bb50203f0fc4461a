mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -x -q -m gpu -k "cln or cLN or causal or composite or assorted" > gpurun_out/r02_c19_pytest.txt 2>&1 || { tail -30 gpurun_out/r02_c19_pytest.txt; exit 1; }
tail -2 gpurun_out/r02_c19_pytest.txt
python benchmarks/cln_only.py 2>&1 | grep -v amdgpu.ids
CONFIG=causal ROUNDS=3 python benchmarks/ab_step.py "composite=1" 2>&1 | grep -v amdgpu.ids
ROUNDS=3 python benchmarks/ab_step.py "composite=1" 2>&1 | grep -v amdgpu.ids
