# Generic PMC pass: bash benchmarks/pmc_kernel.sh <kernel-name-substring> <script> [args]   (GPU box)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
PAT=$1; shift
SCRIPT=$1; shift
TAG=pmc_$(basename $SCRIPT .py)_${1:-x}
rm -rf $R/gpurun_out/${TAG}_*
cd /tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/${TAG}_$i -o p --output-format csv -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -n 5 $R/gpurun_out/${TAG}_$i.log; }
done
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for d in sorted(glob.glob('gpurun_out/${TAG}_*')):
    for f in glob.glob(d + '/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if '$PAT' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(agg.items()):
    print('%-28s %.5g' % (c, sum(v[3:]) / max(1, len(v[3:]))))
PY
