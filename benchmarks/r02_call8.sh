mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wgrad or temporal_block or model" > gpurun_out/r02_c8_pytest.txt 2>&1 || { tail -40 gpurun_out/r02_c8_pytest.txt; exit 1; }
tail -2 gpurun_out/r02_c8_pytest.txt
LAB_CASES="B6,B2" python benchmarks/gemm_lab.py w4 2>&1 | grep -v amdgpu.ids
LAB_CASES="B6,B2" CTN_WGRAD_KERNEL=1 python benchmarks/gemm_lab.py w_old 2>&1 | grep -v amdgpu.ids
for cfg in "CTN_PK_WGS=4" "CTN_PK_WGS=5" "CTN_PK_WGS=6" "CTN_PK_WGS=4 CTN_WGRAD_KERNEL=1" "CTN_PK_WGS=4 CTN_WGRAD_BLOCKS=1024"; do
  echo "== $cfg"; env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-150
done
