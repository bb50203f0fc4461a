# HBM traffic of one kernel from the PMC counters, as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (TCC slots), FETCH_SIZE doubled on gfx950 for wide coalesced reads, WRITE_SIZE exact.
# usage (GPU box): bash benchmarks/pmc_traffic.sh <kernel-name-substring> <script.py> <out.json> [family label] [script args] [arith]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
PAT=$1; SCRIPT=$2; OUT=$3; FAM=${4:-}; SARGS=${5:-}; ARITHNAME=${6:-fp32}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_traffic_$c
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_traffic_$c -o p --output-format csv -- python3 $R/$SCRIPT $SARGS > $R/gpurun_out/pmc_traffic_$c.log 2>&1 || echo "$c pass failed"
done
cd $R
PAT="$PAT" OUT="$OUT" FAM="$FAM" SCRIPT="$SCRIPT" ARITHNAME="$ARITHNAME" python3 - <<'PY'
import csv, glob, collections, json, os
pat = os.environ["PAT"]
agg, name = collections.defaultdict(list), None
for f in glob.glob('gpurun_out/pmc_traffic_*/*counter_collection.csv') + glob.glob('gpurun_out/pmc_traffic_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            name = r['Kernel_Name']
res = {"kernel": name, "family": os.environ["FAM"], "arith": os.environ["ARITHNAME"],
       "command": "bash benchmarks/pmc_traffic.sh '%s' %s  (rocprofv3 --kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE)" % (pat, os.environ["SCRIPT"])}
for c, v in sorted(agg.items()):
    v = v[3:] if len(v) > 6 else v
    res[c + "_KB_mean"] = round(sum(v) / max(1, len(v)), 2)
    res[c + "_launches"] = len(v)
if "FETCH_SIZE_KB_mean" in res and "WRITE_SIZE_KB_mean" in res:
    res["correction"] = "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact"
    res["hbm_bytes_per_launch"] = int(round((2 * res["FETCH_SIZE_KB_mean"] + res["WRITE_SIZE_KB_mean"]) * 1024))
json.dump(res, open(os.environ["OUT"], "w"), indent=1)
print(json.dumps(res, indent=1))
PY
