mkdir -p gpurun_out
export TMPDIR=/tmp; R=$PWD
for cfg in "CTN_PW_KERNEL=1" "CTN_PW_KERNEL=2 CTN_PK_WGS=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  cd /tmp
  env $cfg rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_c9_$tag -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02_c9_$tag.log 2>&1
  cd $R; echo "== $cfg"; python benchmarks/kstats.py gpurun_out/r02_c9_$tag 10 14
done
