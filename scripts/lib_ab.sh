#!/bin/bash
# same-box A / B of the step between two builds of the library: usage A=pointcloududa_amd/lib/libpcuda_hip.so B=... bash scripts/lib_ab.sh [workloads...]
cd "${GRAFT_REPO_ROOT:-.}"
WLS="${@:-full_uda}"
for rep in 1 2 3; do
  for wl in $WLS; do
    for v in "$A" "$B" $C; do
      IFS=: read lib envs <<< "$v"
      env PCUDA_LIB=$lib $envs python bench.py --workload $wl --steps 60 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$wl $v', d['value'], d['ms_per_step'], d['clock_ghz_under_load'])"
    done
  done
done
