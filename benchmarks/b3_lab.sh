# GPU box: time every GEMM form under each ablation build
for form in K1 K3 B1 B5; do
  line="$form: full $(python benchmarks/b3_only.py $form 2>/dev/null | grep -o '[0-9.]* us')"
  for tag in NK1 NOEPI NOCOMPUTE NOSTORE; do
    line="$line | $tag $(CTN_LIB_PATH=benchmarks/lab_b3_$tag.so python benchmarks/b3_only.py $form 2>/dev/null | grep -o '[0-9.]* us')"
  done
  echo "$line"
done
