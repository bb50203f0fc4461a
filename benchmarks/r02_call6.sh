mkdir -p gpurun_out
export TMPDIR=/tmp; R=$PWD
for w in 8 4; do
  cd /tmp
  CTN_PK_WGS=$w rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_c6_prof_w$w -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02_c6_prof_w$w.log 2>&1
  cd $R; echo "== WGS=$w"; python benchmarks/kstats.py gpurun_out/r02_c6_prof_w$w 7 16
done
