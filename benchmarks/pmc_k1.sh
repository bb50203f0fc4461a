export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_k1_$i -o p --output-format csv -- python3 $R/benchmarks/dominant_kernel.py -1 > $R/gpurun_out/pmc_k1_$i.log 2>&1 || { echo "pass $i failed"; tail -n 5 $R/gpurun_out/pmc_k1_$i.log; }
done
cd $R
python benchmarks/pmc_summary.py gpurun_out/pmc_k1_1 gpurun_out/pmc_k1_2 gpurun_out/pmc_k1_3 gpurun_out/pmc_k1_4
