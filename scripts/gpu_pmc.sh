#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE) of single conv layers: ONE counter per rocprofv3 pass (both in one pass
# needed kernel replay and did not finish), no trace domains.   usage: CASES="g64 d4" bash scripts/gpu_pmc.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-r01}
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc/$c -o $TAG -- python3 scripts/conv_micro.py ${CASES:-g64} > gpurun_out/pmc_${c}_$TAG.log 2>&1
  echo "$c pass rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, os
tag = os.environ.get("TAG", "r01")
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"][:64]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("gpurun_out/%s_pmc_traffic.csv" % tag, "w") as o:
    o.write("kernel,launches,FETCH_SIZE_per_launch_KB_raw,WRITE_SIZE_per_launch_KB\n")
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", [0, 1])[0]):
        f_, w_ = d.get("FETCH_SIZE", [0.0, 1]), d.get("WRITE_SIZE", [0.0, 1])
        o.write('"%s",%d,%.1f,%.1f\n' % (k, max(f_[1], w_[1]), f_[0] / max(f_[1], 1), w_[0] / max(w_[1], 1)))
print(open("gpurun_out/%s_pmc_traffic.csv" % tag).read()[:3000])
PY
