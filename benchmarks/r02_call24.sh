ROUNDS=3 python benchmarks/ab_step.py "side=1" "side=0" 2>&1 | grep -v amdgpu.ids
