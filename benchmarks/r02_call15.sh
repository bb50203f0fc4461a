mkdir -p gpurun_out
bash benchmarks/pmc_traffic.sh pw_wgrad4_kernel benchmarks/wgrad_only.py gpurun_out/r02_pmc_dominant_kernel.json "B6 pw_wgrad + slab_reduce (dW1 and the small layers)"
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_c15_prof -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/r02_c15_prof.log 2>&1
cd $R; python benchmarks/kstats.py gpurun_out/r02_c15_prof 7 16
