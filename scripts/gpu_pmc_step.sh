#!/bin/bash
# HBM traffic of the whole benchmark step per kernel: one rocprofv3 --pmc pass per counter (FETCH_SIZE, WRITE_SIZE),
# no trace domains; summary -> gpurun_out/<TAG>_pmc_traffic.csv (copy to profiles/)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-r03}
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  PCUDA_DSTREAMS=0 timeout ${PMC_TIMEOUT:-500} rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc/$c -o $TAG -- python3 bench.py --steps 2 --warmup 1 --settle 2 --no-cpu-baseline --no-roofline > gpurun_out/pmc_${c}_$TAG.log 2>&1
  echo "$c pass rc=$?"; tail -1 gpurun_out/pmc_${c}_$TAG.log | cut -c1-120
done
python3 - <<'PY'
import csv, glob, collections, os
tag = os.environ.get("TAG", "r01")
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        name = name.split("<")[0] if not name.startswith(("igemm", "wgrad_kernel", "wgrad3_kernel")) else name
        a = agg[name[:80]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
rows = []
for k, d in agg.items():
    f_, w_ = d.get("FETCH_SIZE", [0.0, 0]), d.get("WRITE_SIZE", [0.0, 0])
    rows.append((k, max(f_[1], w_[1]), f_[0], w_[0]))
rows.sort(key=lambda r: -(r[2] + r[3]))
import sys
sys.path.insert(0, ".")
from pointcloududa_amd._lib import csrc_hash
with open("gpurun_out/%s_pmc_traffic.csv" % tag, "w") as o:
    o.write("# csrc_sha256=%s\n" % csrc_hash())
    o.write("kernel,launches,FETCH_SIZE_KB_sum_raw,WRITE_SIZE_KB_sum,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch\n")
    for k, n, f_, w_ in rows:
        o.write('"%s",%d,%.0f,%.0f,%.1f,%.1f\n' % (k, n, f_, w_, f_ / max(n, 1), w_ / max(n, 1)))
print(open("gpurun_out/%s_pmc_traffic.csv" % tag).read()[:2500])
PY
