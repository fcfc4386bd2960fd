#!/bin/bash
# SQ / TA / TCP counters of single conv layers (scripts/conv_micro.py cases), one rocprofv3 --pmc pass per counter set,
# no trace domains; aggregates per kernel into gpurun_out/${TAG}_pmc_conv.csv.   usage: CASES="g32" TAG=r02 bash scripts/pmc_conv.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-r03}
rm -rf gpurun_out/pmcc; mkdir -p gpurun_out/pmcc
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum"
 "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
)
# NSETS=2: only the issue / MFMA sets (the per-layer-class table of profiles/rNN_mfma_counters.csv)
i=0
for set in "${SETS[@]:0:${NSETS:-6}}"; do
  timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcc/s$i -o $TAG -- python3 ${SCRIPT:-scripts/conv_micro.py} ${CASES:-g32} > gpurun_out/pmcc_s$i.log 2>&1
  echo "set $i rc=$?"
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections, os, subprocess
tag = os.environ.get("TAG", "r02")
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("gpurun_out/%s_pmc_conv.csv" % tag, "w") as o:
    o.write("kernel,counter,per_launch,launches\n")
    for k, d in sorted(agg.items()):
        if not any(t in k for t in ("igemm", "conv3ap", "conv3rs", "wgrad", "d1_fwd", "d1_dgrad", "d5_fwd", "c1_fwd", "pw_")): continue
        for c, (v, n) in sorted(d.items()):
            o.write('"%s",%s,%.1f,%d\n' % (k[:90], c, v / max(n, 1), n))
print(open("gpurun_out/%s_pmc_conv.csv" % tag).read()[:6000])
PY
