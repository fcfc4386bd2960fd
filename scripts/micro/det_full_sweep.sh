#!/bin/bash
# debug aid: failure rate of scripts/micro/concurrent_det_full.py under each switch (one configuration per argument)
N=${N:-10}
[ $# -eq 0 ] && set -- "X=1"
for cfg in "$@"; do
  bad=0
  for i in $(seq $N); do
    out=$(env $cfg timeout -k 10 170 python scripts/micro/concurrent_det_full.py ${B:-8} 2 2>&1 | grep -E "concurrent processes")
    echo "$out" | grep -q "bit-identical" || bad=$((bad+1))
  done
  echo "== [$cfg] failures $bad / $N"
done
