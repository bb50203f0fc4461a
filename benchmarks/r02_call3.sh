mkdir -p gpurun_out
export LAB_CASES="--t,K1t,B5,B1"
for v in CLOCK NO_BARRIER NO_GLOBAL NO_LDS_READ NO_STORE; do
  CTN_LIB_PATH=$PWD/benchmarks/lab/libctn_$v.so python benchmarks/gemm_lab.py $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r02_c3_lab.txt
cat gpurun_out/r02_c3_lab.txt
