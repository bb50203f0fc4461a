# HBM traffic of the dominant GEMM (K1 shape): FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section).
# usage (GPU box): bash benchmarks/pmc_traffic.sh   -> gpurun_out/pmc_traffic.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_traffic_$c -o p --output-format csv -- python3 $R/benchmarks/dominant_kernel.py -1 > $R/gpurun_out/pmc_traffic_$c.log 2>&1 || echo "$c pass failed"
done
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_traffic_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'pw_gemm' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(agg.items()):
    print('%s mean over launches 4.. : %.2f KB (n=%d)' % (c, sum(v[3:]) / max(1, len(v[3:])), len(v)))
PY
