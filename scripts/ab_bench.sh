# A/B of two libraries on ONE box, alternating: bash scripts/ab_bench.sh <libA> <libB> [steps] [workload]
cd "${GRAFT_REPO_ROOT:-.}"
A=$1; B=$2; STEPS=${3:-60}; WL=${4:-full_uda}
for rep in 1 2; do
  for L in $A $B; do
    PCUDA_LIB=$L python bench.py --steps $STEPS --warmup 20 --workload $WL --no-cpu-baseline --no-roofline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$L', d['value'], d['ms_per_step'], d['clock_ghz_under_load'])"
  done
done
