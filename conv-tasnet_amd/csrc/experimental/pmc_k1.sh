# PMC stall / LDS counters of the dominant GEMM (K1 shape).  usage: bash benchmarks/pmc_k1.sh [fp32|x6] [tile]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
export CTN_GEMM_MODE=${1:-fp32}
TILE=${2:--1}
TAG=pmc_k1_${CTN_GEMM_MODE}_t${TILE}
cd /tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/${TAG}_$i -o p --output-format csv -- python3 $R/benchmarks/dominant_kernel.py $TILE > $R/gpurun_out/${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -n 5 $R/gpurun_out/${TAG}_$i.log; }
done
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for d in sorted(glob.glob('gpurun_out/${TAG}_?')):
    for f in glob.glob(d + '/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'pw_gemm' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(agg.items()):
    print('%-28s %.5g' % (c, sum(v[3:]) / max(1, len(v[3:]))))
PY
