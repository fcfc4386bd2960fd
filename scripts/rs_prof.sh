#!/bin/bash
# kernel durations of the row-streaming kernel against the kernel it replaces (rocprofv3 --kernel-trace --stats of scripts/conv_micro.py):
# the micro-benchmark's wall time per call is host-bound below ~0.17 ms.   usage: bash scripts/rs_prof.sh [cases...] > gpurun_out/rs_prof.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
CASES="${@:-g32}"
for rs in 0 1; do
  rm -rf /tmp/rsprof$rs
  PCUDA_RS=$rs MICRO_REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rsprof$rs -o p -- python3 scripts/conv_micro.py $CASES > /dev/null 2>&1
  echo "== PCUDA_RS=$rs"
  python3 - /tmp/rsprof$rs <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('conv3rs', 'igemm', 'wgrad')): print('%8.1f us x %4s  %s' % (float(r['AverageNs']) / 1e3, r['Calls'], n[:100]))
P
done
