# GPU box: time every GEMM form under each ablation build (benchmarks/gemm_ablate_build.sh)
TAGS="${TAGS:-NK1 NOEPI NOMFMA NOA NOSPLIT NOLDSRD}"
for form in ${FORMS:-K1 K3 B1 B5}; do
  line="$form: full $(python benchmarks/gemm_only.py $form 2>/dev/null | grep -o '[0-9.]* us')"
  for tag in $TAGS; do
    line="$line | $tag $(CTN_LIB_PATH=benchmarks/lab_gemm_$tag.so python benchmarks/gemm_only.py $form 2>/dev/null | grep -o '[0-9.]* us')"
  done
  echo "$line"
done
