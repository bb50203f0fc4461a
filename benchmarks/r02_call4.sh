mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02_c4_pytest.txt 2>&1 || { tail -40 gpurun_out/r02_c4_pytest.txt; exit 1; }
tail -3 gpurun_out/r02_c4_pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_c4_bench.txt 2>&1; tail -1 gpurun_out/r02_c4_bench.txt | cut -c1-200
CTN_PW_KERNEL=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_c4_bench_old.txt 2>&1; tail -1 gpurun_out/r02_c4_bench_old.txt | cut -c1-200
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_c4_prof -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02_c4_prof.log 2>&1
cd $R; python benchmarks/kstats.py gpurun_out/r02_c4_prof 2>/dev/null | head -30 || ls gpurun_out/r02_c4_prof
