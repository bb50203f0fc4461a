mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r02_g_pytest.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r02_g_pytest.txt
ROUNDS=2 python benchmarks/ab_step.py "arith=1" 2>&1 | grep -v amdgpu.ids
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
