mkdir -p gpurun_out
bash benchmarks/pmc_traffic.sh 'pw_gemm_b3p_kernel' benchmarks/b3_only.py gpurun_out/r02_pmc_b3_K1.json "K1 pw_gemm<EPI_PRELU_STATS> (1x1 B->H + PReLU/gLN statistics)" K1 b3 | grep -E "hbm_bytes"
bash benchmarks/pmc_traffic.sh 'pw_gemm_b3p_kernel' benchmarks/b3_only.py gpurun_out/r02_pmc_b3_K3.json "K3 pw_gemm<PRO_PRELU_NORM,EPI_RESIDUAL> (1x1 H->B, gLN prologue + residual)" K3 b3 | grep -E "hbm_bytes"
bash benchmarks/pmc_traffic.sh 'pw_gemm_b3p_kernel' benchmarks/b3_only.py gpurun_out/r02_pmc_b3_B1.json "B1 pw_gemm<T,EPI_GLN_BWD> (input gradient W2^T.dout + gLN backward sums)" B1 b3 | grep -E "hbm_bytes"
bash benchmarks/pmc_traffic.sh 'pw_gemm_b3p_kernel' benchmarks/b3_only.py gpurun_out/r02_pmc_b3_B5.json "B5 pw_gemm<T,EPI_RESIDUAL> (input gradient W1^T.dh1 + dout)" B5 b3 | grep -E "hbm_bytes"
