#!/bin/bash
# debug aid: failure rate of tests/test_dist_gpu.py's two-rank comparison (scripts/micro/dp2_diff.py) on one box;
# usage: N=20 scripts/micro/dp2_sweep.sh ["ENV=1 ENV2=0" ...]   (one configuration per argument; none = defaults)
N=${N:-12}
[ $# -eq 0 ] && set -- "X=1"
for cfg in "$@"; do
  bad=0
  for i in $(seq $N); do
    out=$(env $cfg python scripts/micro/dp2_diff.py 2>&1 | grep -E "^(seg|d1|d2|d4) max")
    if [ -z "$out" ] || echo "$out" | grep -qv "max diff vs single 0.0 n diff 0 "; then bad=$((bad+1)); echo "$out" | head -2; fi
  done
  echo "== [$cfg] failures $bad / $N"
done
