#!/bin/bash
# SQ counters of the anti-phase micro (one shape): MFMA pipe busy, waits, LDS conflicts
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
B=scripts/micro/bin/conv3ap_micro
SHAPE=${SHAPE:-"32 256 256 32 32"}
mkdir -p gpurun_out/appmc
for mode in "0 0" "4 2"; do
  set -- $mode
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d gpurun_out/appmc/d$1_p$i -- $B $SHAPE 20 $1 $2 > gpurun_out/appmc/d$1_p$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in ("d0", "d4"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob('gpurun_out/appmc/%s_p*/**/*counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv3ap' not in r['Kernel_Name']: continue
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print('==', d)
    per = {c: agg[c] / n[c] for c in agg}
    for c in sorted(per): print('   %-30s %.4g' % (c, per[c]))
    if 'SQ_BUSY_CYCLES' in per:
        print('   mfma pipe busy = %.3f' % (per['SQ_VALU_MFMA_BUSY_CYCLES'] / (32.0 * per['SQ_BUSY_CYCLES'])))
        wc = per['SQ_WAVE_CYCLES']
        print('   wait_any %.3f wait_inst_any %.3f active %.3f' % (per['SQ_WAIT_ANY'] / wc, per['SQ_WAIT_INST_ANY'] / wc, per['SQ_ACTIVE_INST_ANY'] / wc))
PY
