# Fresh-process A/B of an environment switch in the bench's own regime: bench.py (no CPU leg, no side records) alternately with
# VAR=A and VAR=B, N rounds.  usage: bash benchmarks/ab_env_bench.sh VAR A B [rounds] [bench args...]
V=$1; A=$2; B=$3; N=${4:-3}; shift 4
for r in $(seq $N); do
  for x in $A $B; do
    env $V=$x python bench.py --no-cpu-baseline --no-side-configs --no-side-arith --no-roofline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x', j['value'], j['ms_per_step'])"
  done
done
