mkdir -p gpurun_out
bash benchmarks/pmc_traffic.sh "pw_wgrad4_kernel<0" benchmarks/wgrad_only.py gpurun_out/r02_pmc_wgrad_dW1.json "B6 pw_wgrad + slab_reduce (dW1 = dh1 . x^T)" > /dev/null
bash benchmarks/pmc_traffic.sh "pw_wgrad4_kernel<1" benchmarks/wgrad_only.py gpurun_out/r02_pmc_wgrad_dW2_pro.json "B2 pw_wgrad<PRO> + slab_reduce (dW2 = dout . gLN2(prelu(d))^T)" pro > /dev/null
cat gpurun_out/r02_pmc_wgrad_dW1.json gpurun_out/r02_pmc_wgrad_dW2_pro.json
