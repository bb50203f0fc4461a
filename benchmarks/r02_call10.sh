mkdir -p gpurun_out
( CTN_PW_KERNEL=1 python bench.py --steps 400 --warmup 5 --no-cpu-baseline > gpurun_out/r02_c10_bench.txt 2>&1 ) &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|Temperature (Sensor junction)\|fclk" | head -8; echo --; sleep 0.7; done > gpurun_out/r02_c10_smi.txt
wait $BP
cat gpurun_out/r02_c10_smi.txt | head -60; tail -1 gpurun_out/r02_c10_bench.txt | cut -c1-160
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
