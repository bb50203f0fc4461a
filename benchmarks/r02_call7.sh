mkdir -p gpurun_out
export LAB_CASES="--t,K1t,B5,B1,K3t"
export CTN_LIB_PATH=$PWD/benchmarks/lab/libctn_CLOCK.so
( python benchmarks/gemm_lab.py wt1_w8
for w in 1 2 3; do CTN_PK_WT=2 CTN_PK_WGS=$w python benchmarks/gemm_lab.py wt2_w$w; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_c7_lab.txt
cat gpurun_out/r02_c7_lab.txt
