# usage (GPU box): bash benchmarks/power_lab.sh mfma|step  -> gpurun_out/r02_power_lab[_mfma].txt  (W, us and mJ per launch per case)
# the sampler runs in a shell loop started BEFORE the measured process touches the GPU
mkdir -p gpurun_out
MODE=${1:-kernels}
( while true; do echo "$(date +%s.%N) $(rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$')"; sleep 0.15; done ) > gpurun_out/power_samples.txt &
SP=$!
if [ "$MODE" = mfma ]; then ./benchmarks/mfma_probe.bin 5 > gpurun_out/power_cases.txt 2> gpurun_out/power_lab.err
elif [ "$MODE" = step ]; then python benchmarks/power_lab_step.py > gpurun_out/power_cases.txt 2> gpurun_out/power_lab.err
else echo "usage: power_lab.sh mfma|step   (per-kernel energies: benchmarks/power_lab_gemm.sh)"; kill $SP; exit 1; fi
kill $SP
export MODE
python - <<'PY'
samples = []
for l in open("gpurun_out/power_samples.txt"):
    p = l.split()
    if len(p) == 2:
        samples.append((float(p[0]), float(p[1])))
import os
out = open("gpurun_out/power_lab%s.txt" % {"mfma": "_mfma", "x": "_x", "step": "_step"}.get(os.environ.get("MODE"), ""), "w")
for l in open("gpurun_out/power_cases.txt"):
    p = l.split()
    if p and p[0] == "case":
        name, t0, t1, n, us = p[1], float(p[2]), float(p[3]), int(p[4]), float(p[5])
        extra = l.split("#", 1)[1].strip() if "#" in l else ""
        w = [v for t, v in samples if t0 + 0.5 < t < t1 - 0.2]
        if w:
            pw = sum(w) / len(w)
            line = "%-28s %7.1f W (%2d samples, max %6.1f)  %9.2f us/launch  %8.2f mJ/launch" % (name, pw, len(w), max(w), us, pw * us * 1e-3) + ("  " + extra if extra else "")
        else:
            line = "%-28s no power samples  %9.2f us/launch" % (name, us)
        print(line); out.write(line + "\n")
PY
