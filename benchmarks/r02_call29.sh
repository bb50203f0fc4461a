for form in K1 K3 B1 B5 W1 W2; do
  echo "$form: default $(python benchmarks/b3_only.py $form 2>/dev/null | grep -o '[0-9.]* us') | no-slp $(CTN_LIB_PATH=benchmarks/lab_noslp.so python benchmarks/b3_only.py $form 2>/dev/null | grep -o '[0-9.]* us')"
done
ROUNDS=2 python benchmarks/ab_step.py "arith=1" 2>&1 | grep -v amdgpu.ids
CTN_LIB_PATH=benchmarks/lab_noslp.so ROUNDS=2 python benchmarks/ab_step.py "arith=1" 2>&1 | grep -v amdgpu.ids
