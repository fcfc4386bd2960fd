#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per requested byte for the load forms of the convolution kernels (scripts/micro/fetch_calib.hip);
# writes gpurun_out/${TAG}_fetch_calib.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-r03}
# (the binary is not tracked: build it here, before anything touches the GPU)
mkdir -p scripts/micro/bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/fetch_calib.hip -o scripts/micro/bin/fetch_calib || exit 1
rm -rf gpurun_out/fcal; mkdir -p gpurun_out/fcal
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/fcal/$c -o cal -- ./scripts/micro/bin/fetch_calib > gpurun_out/fcal/$c.log 2>&1
  echo "$c rc=$?"; tail -1 gpurun_out/fcal/$c.log
done
python3 - <<'PY' | tee gpurun_out/${TAG:-r03}_fetch_calib.txt
import csv, glob, collections
req = {"k_x4": 1 << 30, "k_x1": 1 << 30, "k_buf1": 1 << 30, "k_rows": (1 << 20) * 160, "k_w4": 1 << 30}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/fcal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k in req:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]) * 1024.0)      # the counters are in KB
print("kernel      requested_MB  FETCH_SIZE_MB (x = per requested byte)   WRITE_SIZE_MB")
for k in ("k_x4", "k_x1", "k_buf1", "k_rows", "k_w4"):
    f = agg[k].get("FETCH_SIZE", [0.0]); w = agg[k].get("WRITE_SIZE", [0.0])
    fm, wm = sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)
    print("%-10s %10.1f   %10.1f (x %.3f)   %10.1f (x %.3f)" % (k, req[k] / 1e6, fm / 1e6, fm / req[k], wm / 1e6, wm / req[k]))
print("k_rows touches 3 lines of 128 B per 160 requested bytes: lines / requested = 2.4")
PY
