# builds the phase-ablation variants of the split-bf16 GEMM kernels next to the product library (local, cross-compile):
# benchmarks/lab_gemm_<tag>.so   (the hooks are #ifdef CTN_EXP_B3_<tag> in csrc/ctn_gemm_b3.h; never defined in the product build)
set -e
cd "$(dirname "$0")/.."
TAGS="${TAGS:-NK1 NOEPI NOMFMA NOA NOSPLIT NOLDSRD}"   # also: TIMELINE (benchmarks/gemm_timeline.py, ws_timeline.py)
mkdir -p /tmp/lab_objs
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"
python conv-tasnet_amd/_build.py > /dev/null
OBJS=$(ls conv-tasnet_amd/csrc/build/*.o | grep -v ctn_gemm.o)
for tag in $TAGS; do
  defs=$(echo $tag | sed 's/+/ -DCTN_EXP_B3_/g')          # "A+B" builds with both hooks
  ( $HIPCC $FLAGS -DCTN_EXP_B3_$defs $EXTRA_DEFS -c conv-tasnet_amd/csrc/ctn_gemm.hip -o /tmp/lab_objs/ctn_gemm_$tag.o && \
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o benchmarks/lab_gemm_$tag.so $OBJS /tmp/lab_objs/ctn_gemm_$tag.o ) &
done
wait
ls -la benchmarks/lab_gemm_*.so
