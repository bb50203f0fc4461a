mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r02_c18_pytest.txt 2>&1 || { tail -40 gpurun_out/r02_c18_pytest.txt; exit 1; }
tail -2 gpurun_out/r02_c18_pytest.txt
CONFIG=causal ROUNDS=3 python benchmarks/ab_step.py "composite=1" "composite=0" 2>&1 | grep -v amdgpu.ids
python bench.py --config causal --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-420
