#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE) of single conv layers, PMC pass on its own (no trace domains).
# usage: CASES="g64 d4" bash scripts/gpu_pmc.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-r01}
mkdir -p gpurun_out/pmc
timeout 600 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/pmc -o $TAG -- python3 scripts/conv_micro.py ${CASES:-g64} > gpurun_out/pmc_micro_$TAG.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
tag = os.environ.get("TAG", "r01")
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob("gpurun_out/pmc/**/*counter_collection.csv", recursive=True):
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        disp[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for (d, k), c in disp.items():
        a = agg[k[:70]]; a[0] += c.get("FETCH_SIZE", 0.0); a[1] += c.get("WRITE_SIZE", 0.0); a[2] += 1
with open("gpurun_out/%s_pmc_traffic.csv" % tag, "w") as o:
    o.write("kernel,launches,fetch_size_per_launch_kb,write_size_per_launch_kb\n")
    for k, (f_, w_, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        o.write("\"%s\",%d,%.1f,%.1f\n" % (k, n, f_ / n, w_ / n))
print(open("gpurun_out/%s_pmc_traffic.csv" % tag).read()[:2500])
PY
