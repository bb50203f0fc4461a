rm -f gpurun_out/wg_abl2.log
for t in BASE WGNOLOAD WGNOCOMPUTE WGNOCOMPUTE+WGNOLOAD WGNOCOMPUTE+WGNOSPLIT+WGNOWRITE WGNOSPLIT+WGNOWRITE WGNOLOAD+WGNOSPLIT+WGNOWRITE WGSGB WGSGB+WGNOLOAD; do
  if [ $t = BASE ]; then lib=conv-tasnet_amd/libctn_hip.so; else lib=benchmarks/lab_gemm_$t.so; fi
  echo "$t" >> gpurun_out/wg_abl2.log
  CTN_LIB_PATH="$lib" FORMS="W1 W2" SECONDS_PER_CASE=0.4 python benchmarks/gemm_lab.py b3_wgrad_blocks=256 2>&1 | grep b3_wgrad >> gpurun_out/wg_abl2.log || exit 1
done
cat gpurun_out/wg_abl2.log
