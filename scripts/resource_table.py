"""kernel resource table from a `hipcc -Rpass-analysis=kernel-resource-usage` log: VGPRs, scratch, spills per kernel.
usage: python scripts/resource_table.py <log> [substring ...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
names, rows = [], []
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    names.append(b.split()[0])
    rows.append((g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"), g("SGPRs Spill"), g("TotalSGPRs")))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print("%-100s vgpr agpr scratch vspill sspill sgpr" % "kernel")
for n, r in zip(dem, rows):
    n = n.replace("void ", "").replace("(IgemmParams, int, int)", "").replace("(IgemmParams, int)", "").replace("(WgradParams)", "")
    if all(s in n for s in sys.argv[2:]):
        print("%-100s %4d %4d %7d %6d %6d %4d" % ((n,) + r))
