mkdir -p gpurun_out
for cfg in "CTN_PW_KERNEL=2" "CTN_PW_KERNEL=1" "CTN_PW_KERNEL=2 CTN_SIDE_STREAM=0" "CTN_PW_KERNEL=1 CTN_SIDE_STREAM=0" "CTN_PW_KERNEL=2 CTN_PK_WGS=4" "CTN_PW_KERNEL=2 CTN_PK_WGS=2"; do
  echo "== $cfg"; env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-150
done > gpurun_out/r02_c5.txt 2>&1
cat gpurun_out/r02_c5.txt
