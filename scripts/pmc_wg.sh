cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
CASE=${CASE:-g64}
mkdir -p gpurun_out/pmc
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc/p$i -- python3 scripts/conv_micro.py $CASE > gpurun_out/pmc/p$i.log 2>&1
  tail -2 gpurun_out/pmc/p$i.log
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'wgrad' not in k and 'igemm' not in k: continue
    print(k)
    for c, v in sorted(d.items()): print('   %-34s %.4g' % (c, v))
PY
