# usage (GPU box): bash benchmarks/power_lab_gemm.sh <out.txt>   -- W, us and mJ per launch: product library under h3 / b6 / fp32 and
# every ablation build present (benchmarks/lab_gemm_<TAG>.so), FORMS env selects the GEMM forms
OUT=${1:-gpurun_out/power_lab_b6.txt}
mkdir -p gpurun_out
( while true; do echo "$(date +%s.%N) $(rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$')"; sleep 0.12; done ) > gpurun_out/power_samples.txt &
SP=$!
: > gpurun_out/power_cases.txt
for ar in ${ARITHS:-h3 b6 fp32}; do LABEL=$ar ARITH=$ar python benchmarks/power_lab_gemm.py >> gpurun_out/power_cases.txt 2>> gpurun_out/power_lab.err; done
for t in ${TILES:-}; do LABEL=b6_tile$t ARITH=b6 TILE=$t FORMS="K1 K3 B1 B5" python benchmarks/power_lab_gemm.py >> gpurun_out/power_cases.txt 2>> gpurun_out/power_lab.err; done
for tag in ${TAGS:-NK1 NOEPI NOMFMA NOA NOSPLIT NOLDSRD}; do
  [ -f benchmarks/lab_gemm_$tag.so ] && LABEL=${LABARITH:-b6}_$tag ARITH=${LABARITH:-b6} CTN_LIB_PATH=benchmarks/lab_gemm_$tag.so FORMS="${LABFORMS:-K1 B1}" python benchmarks/power_lab_gemm.py >> gpurun_out/power_cases.txt 2>> gpurun_out/power_lab.err
done
kill $SP
OUT=$OUT python - <<'PY'
import os
samples = []
for l in open("gpurun_out/power_samples.txt"):
    p = l.split()
    if len(p) == 2:
        samples.append((float(p[0]), float(p[1])))
out = open(os.environ["OUT"], "w")
for l in open("gpurun_out/power_cases.txt"):
    p = l.split()
    if p and p[0] == "case":
        name, t0, t1, n, us = p[1], float(p[2]), float(p[3]), int(p[4]), float(p[5])
        w = [v for t, v in samples if t0 + 0.4 < t < t1 - 0.15]
        if w:
            pw = sum(w) / len(w)
            line = "%-22s %7.1f W (%2d samples, max %6.1f)  %8.2f us/launch  %7.2f mJ/launch" % (name, pw, len(w), max(w), us, pw * us * 1e-3)
        else:
            line = "%-22s no power samples  %8.2f us/launch" % (name, us)
        print(line); out.write(line + "\n")
PY
