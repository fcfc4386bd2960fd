#!/bin/bash
# Per-LAYER HBM traffic of the conv kernels (forward, dgrad, wgrad of every scripts/conv_micro.py case): FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes (they do not fit one), no trace domains; dispatches are mapped back to
# (case, op) by their order.  Writes gpurun_out/${TAG}_pmc_layers.csv.   usage: TAG=r02 bash scripts/pmc_layers.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-r03}
CASES=${CASES:-"g32 g64 g128 g256 b512 d2 d4"}
rm -rf gpurun_out/pmcl; mkdir -p gpurun_out/pmcl
for c in FETCH_SIZE WRITE_SIZE; do
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmcl/$c -o $TAG -- python3 scripts/conv_micro.py $CASES > gpurun_out/pmcl_$c.log 2>&1
  echo "$c pass rc=$?"
done
CASES="$CASES" python3 - <<'PY'
import csv, glob, os, re, collections
tag = os.environ.get("TAG", "r02"); cases = os.environ["CASES"].split()
# conv_micro.py per case: forward x11, dgrad x11 (x parity classes), wgrad x11 (+ reduce), in that order
GEOM = {"g32": (32, 32, 32, 256, 256, 3, 1), "g64": (32, 64, 64, 128, 128, 3, 1), "g128": (32, 128, 128, 64, 64, 3, 1),
        "g256": (32, 256, 256, 32, 32, 3, 1), "b512": (32, 512, 512, 16, 16, 3, 1), "d2": (32, 64, 128, 129, 129, 4, 2),
        "d4": (32, 256, 512, 33, 33, 4, 2)}
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/pmcl/%s/**/*counter_collection.csv" % ctr, recursive=True)
    if not fs: continue
    recs = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Dispatch_Id"]))
    recs = [r for r in recs if r["Counter_Name"] == ctr and ("igemm" in r["Kernel_Name"] or "conv3ap_kernel" in r["Kernel_Name"] or "conv3rs_kernel" in r["Kernel_Name"] or "wgrad_kernel" in r["Kernel_Name"] or "wgrad3_kernel" in r["Kernel_Name"] or "wgrad3r_kernel" in r["Kernel_Name"])]
    # conv dispatches arrive in a fixed order: per case 11 forward launches, 11 x dgrad launches per call (stride 1: one;
    # k = 4 / stride 2 / pad 2: one per ROW parity, the column classes paired -- four with PCUDA_DGRAD_PAIR=0), 11
    # weight-gradient launches (their reduce kernels are filtered out above)
    pos = 0
    pair = os.environ.get("PCUDA_DGRAD_PAIR", "1") != "0"
    for ci, cs in enumerate(cases):
        st = GEOM[cs][6]
        for op, cnt in enumerate((11, 11 * (st * st if (st == 1 or not pair) else st), 11)):
            for r in recs[pos:pos + cnt]:
                assert ("wgrad_kernel" in r["Kernel_Name"] or "wgrad3_kernel" in r["Kernel_Name"] or "wgrad3r_kernel" in r["Kernel_Name"]) == (op == 2), (cs, op, r["Kernel_Name"])
                rows[(ci, op)][ctr].append(float(r["Counter_Value"]))
            pos += cnt
    assert pos == len(recs), (pos, len(recs))
with open("gpurun_out/%s_pmc_layers.csv" % tag, "w") as o:
    o.write("# FETCH_SIZE counts 64 B per 128-byte line touched, for every load form of these kernels (dwordx4, dword, buffer dword: profiles/r03_fetch_calib.txt): fabric bytes = 2 x FETCH_SIZE + WRITE_SIZE\n")
    o.write("case,op,n,cin,cout,h,w,k,stride,launches,FETCH_SIZE_KB_raw_per_launch,WRITE_SIZE_KB_per_launch,hbm_MB_per_launch(2xFETCH+WRITE),algorithmic_MB\n")
    for (ci, op), d in sorted(rows.items()):
        n, cin, cout, h, w, k, s = GEOM[cases[ci]]
        oh, ow = (h + 2 * (k // 2 if s == 1 else 2) - k) // s + 1, (w + 2 * (k // 2 if s == 1 else 2) - k) // s + 1
        xin, yout = n * cin * h * w * 4 / 1e6, n * cout * oh * ow * 4 / 1e6
        alg = xin + yout
        f, wv = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
        # per CALL of the op: conv_micro runs every op 11 times; a dgrad of stride 2 is 2 (paired) or 4 kernels per call
        per = lambda v: sum(v) / 11.0
        o.write("%s,%s,%d,%d,%d,%d,%d,%d,%d,%d,%.1f,%.1f,%.1f,%.1f\n" % (cases[ci], ("fwd", "dgrad", "wgrad")[op], n, cin, cout, h, w, k, s,
                len(f), per(f), per(wv), (2 * per(f) + per(wv)) / 1e3, alg))
print(open("gpurun_out/%s_pmc_layers.csv" % tag).read())
PY
