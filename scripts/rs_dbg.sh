#!/bin/bash
# switched-off-phase bounds of the row-streaming kernel (make BUILD=build_rsdbg OUT=../lib/libpcuda_rsdbg.so XFLAGS=-DPCUDA_RS_DEBUG):
# PCUDA_RSDBG bits: 1 no MFMAs, 2 no stores, 4 no input loads, 8 no conversion; kernel durations from rocprofv3 (the micro-benchmark's
# wall time per call is host-bound below ~0.17 ms).   usage: bash scripts/rs_dbg.sh > gpurun_out/rs_dbg.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for d in ${DBGS:-0 1 2 4 8 6 14 15 7}; do
  rm -rf /tmp/rsdbg
  PCUDA_LIB=pointcloududa_amd/lib/libpcuda_rsdbg.so PCUDA_RSDBG=$d MICRO_REPS=30 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rsdbg -o p -- python3 scripts/conv_micro.py ${CASE:-g32} > /dev/null 2>&1
  python3 - /tmp/rsdbg $d <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'conv3rs_kernel<1' in n: print('PCUDA_RSDBG=%-3s %8.1f us x %4s  %s' % (sys.argv[2], float(r['AverageNs']) / 1e3, r['Calls'], n[:90]))
P
done
