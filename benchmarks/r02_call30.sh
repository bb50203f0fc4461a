ROUNDS=4 python benchmarks/ab_step.py "fin_side=1" "fin_side=0" 2>&1 | grep -v amdgpu.ids
CONFIG=causal ROUNDS=4 python benchmarks/ab_step.py "fin_side=1" "fin_side=0" 2>&1 | grep -v amdgpu.ids
