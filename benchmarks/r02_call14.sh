mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma16 or persistent" > gpurun_out/r02_c14_pytest.txt 2>&1 || { tail -40 gpurun_out/r02_c14_pytest.txt; exit 1; }
tail -2 gpurun_out/r02_c14_pytest.txt
python benchmarks/ab_step.py "pw_tile=3,wgrad_mf=32" "pw_tile=11,wgrad_mf=32" "pw_tile=3,wgrad_mf=16" "pw_tile=11,wgrad_mf=16" "pw_tile=11,wgrad_mf=16,block_wt=0" "pw_tile=3,wgrad_mf=32,block_wt=0" 2>&1 | grep -v amdgpu.ids
