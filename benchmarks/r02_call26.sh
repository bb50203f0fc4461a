mkdir -p gpurun_out
bash benchmarks/power_lab.sh step
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench_b3 -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_bench_b3.log 2>&1
cd $R
head -14 gpurun_out/prof_bench_b3/p_kernel_stats.csv | cut -c1-150
bash benchmarks/pmc_traffic.sh 'dw_bwd_kernel' benchmarks/dw_bwd_only.py gpurun_out/r02_pmc_dw_bwd.json "B3 dw_bwd fused (gLN2'.PReLU2'.dw^T)" "" any | grep -E "hbm_bytes|kernel"
