export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_causal_tl -o p --output-format csv -- python3 $R/bench.py --config causal --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-side-arith --no-side-configs > $R/gpurun_out/prof_causal_tl.log 2>&1
cd $R
F=$(ls gpurun_out/prof_causal_tl/*kernel_trace.csv gpurun_out/prof_causal_tl/*/*kernel_trace.csv 2>/dev/null | head -1)
python benchmarks/tools/timeline.py $F 2 0 2000 > gpurun_out/causal_timeline.txt
wc -l gpurun_out/causal_timeline.txt
rm -rf gpurun_out/prof_causal_tl
