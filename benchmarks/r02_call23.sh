mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
for f in W1 W2; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b3_$f -o p --output-format csv -- python3 $R/benchmarks/b3_only.py $f 40 > $R/gpurun_out/prof_b3_$f.log 2>&1
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_b3_$f/p_kernel_stats.csv")))
for r in rows[:6]:
    print("$f", r["Name"][:90], r["Calls"], r["AverageNs"])
PY
done
