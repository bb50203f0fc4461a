# builds the phase-ablation variants of the b3 kernels next to the product library (local, cross-compile): benchmarks/lab_b3_<tag>.so
set -e
cd "$(dirname "$0")/.."
for tag in NK1 NOEPI NOCOMPUTE NOSTORE; do
  CTN_EXTRA_HIPCC_FLAGS=-DCTN_EXP_B3_$tag python conv-tasnet_amd/_build.py --force > /dev/null 2>&1
  cp conv-tasnet_amd/libctn_hip.so benchmarks/lab_b3_$tag.so
done
python conv-tasnet_amd/_build.py --force > /dev/null 2>&1
ls -la benchmarks/lab_b3_*.so
