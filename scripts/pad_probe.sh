#!/bin/bash
# do power-of-two plane strides cost the full-resolution kernels bandwidth?  kernel durations (rocprofv3) of scripts/conv_micro.py with
# every activation plane's pitch padded by MICRO_PAD floats.   usage: bash scripts/pad_probe.sh [cases...] > gpurun_out/pad_probe.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
CASES="${@:-g32 g64}"
for pad in 0 64 0 64 256 1056; do
  rm -rf /tmp/padp
  MICRO_PAD=$pad MICRO_REPS=30 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/padp -o p -- python3 scripts/conv_micro.py $CASES > /tmp/padp.log 2>&1
  echo "== MICRO_PAD=$pad"; grep -v amdgpu.ids /tmp/padp.log | cut -c1-110
  python3 - /tmp/padp <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('conv3rs', 'conv3ap', 'igemm', 'wgrad')) and 'reduce' not in n: print('%8.1f us x %4s  %s' % (float(r['AverageNs']) / 1e3, r['Calls'], n.replace('(anonymous namespace)::', '')[:90]))
P
done
