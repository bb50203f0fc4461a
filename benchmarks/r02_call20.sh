mkdir -p gpurun_out
python benchmarks/b3_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_b3_check.txt
ROUNDS=3 python benchmarks/ab_step.py "arith=0" "arith=1" "arith=1,b3_tile=1" "arith=1,b3_tile=3" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_b3_ab.txt
