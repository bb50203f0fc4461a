# rocprofv3 kernel statistics of the bench step under two arithmetics, side by side.  usage: gpurun -- bash benchmarks/prof_arith.sh h3 b3 [tag]
A1=${1:-h3}; A2=${2:-b3}; T=${3:-prof}
export TMPDIR=/tmp
R=$(pwd)
mkdir -p $R/gpurun_out
for A in $A1 $A2; do
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_$A -o p --output-format csv -- python3 $R/bench.py --arith $A --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-side-arith > $R/gpurun_out/${T}_$A.log 2>&1 || { tail -20 $R/gpurun_out/${T}_$A.log; exit 1; }
  cd $R
  cp $(ls gpurun_out/${T}_$A/*kernel_stats.csv gpurun_out/${T}_$A/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_${A}_kernel_stats.csv
  echo "== $A"; tail -1 gpurun_out/${T}_$A.log | cut -c1-200
  python benchmarks/kstats.py gpurun_out/${T}_$A 7 24
done
