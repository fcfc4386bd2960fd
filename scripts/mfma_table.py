"""profiles/rNN_mfma_counters.csv from gpurun_out/rNN_pmc_conv.csv: per kernel (= layer class of scripts/conv_micro.py)
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 (the counter
sums the 32 shader engines); issue mix per launch."""
import csv, collections, sys
src, dst = sys.argv[1], sys.argv[2]
d = collections.defaultdict(dict)
for r in csv.DictReader(open(src)):
    d[r["kernel"]][r["counter"]] = float(r["per_launch"])
with open(dst, "w") as o:
    o.write("kernel,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES,mfma_util,SQ_WAIT_ANY_frac,SQ_WAIT_INST_ANY_frac,"
            "SQ_ACTIVE_INST_VALU_frac,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_MFMA,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,valu_per_mfma\n")
    for k, c in sorted(d.items()):
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        g = lambda n: c.get(n, 0.0)
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
        o.write('"%s",%.0f,%.0f,%.0f,%.4f,%.4f,%.4f,%.4f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.2f\n' % (
            k, wc, g("SQ_BUSY_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES") / max(32.0 * g("SQ_BUSY_CYCLES"), 1.0),
            g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"),
            g("SQ_INSTS_MFMA"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"),
            g("SQ_INSTS_VALU") / max(g("SQ_INSTS_MFMA"), 1.0)))
print(open(dst).read()[:3000])
