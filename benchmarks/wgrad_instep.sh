# in-step sensitivity of the training step to the weight-gradient kernel: product library against lab builds (one process each)
rm -f gpurun_out/wg_instep.log
for t in ${LIBS:-BASE WGSGB WGNOCOMPUTE WGNULL BASE}; do
  if [ $t = BASE ]; then lib=conv-tasnet_amd/libctn_hip.so; else lib=benchmarks/lab_gemm_$t.so; fi
  echo "$t" >> gpurun_out/wg_instep.log
  CTN_LIB_PATH="$lib" ROUNDS=3 STEPS=10 python benchmarks/ab_step.py "side=1" 2>&1 | grep median >> gpurun_out/wg_instep.log || exit 1
done
cat gpurun_out/wg_instep.log
