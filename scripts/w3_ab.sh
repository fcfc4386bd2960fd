#!/bin/bash
# wgrad3_kernel durations (rocprofv3) between two builds of the library: usage A=... B=... bash scripts/w3_ab.sh [cases]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
CASES="${@:-g128 g256}"
for lib in $A $B $A $B; do
  rm -rf /tmp/w3ab
  PCUDA_LIB=$lib MICRO_REPS=30 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w3ab -o p -- python3 scripts/conv_micro.py $CASES > /dev/null 2>&1
  echo "== $lib"
  python3 - /tmp/w3ab <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'wgrad3_kernel' in r['Name']: print('%8.1f us x %4s  %s' % (float(r['AverageNs']) / 1e3, r['Calls'], r['Name'][:80]))
P
done
