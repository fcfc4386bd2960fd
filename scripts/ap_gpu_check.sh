#!/bin/bash
# round 6: the anti-phase kernel in the product: its kernel-level tests, the convolution / network / step suites, then the
# benchmark with and without it on the same box
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_conv_ap_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/ap_tests.txt || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee -a gpurun_out/ap_tests.txt
for i in 1 2; do
  PCUDA_AP=0 timeout -k 10 200 python bench.py --steps 60 --warmup 20 2>/dev/null | tail -1 > gpurun_out/ap_bench_off_$i.json
  PCUDA_AP=1 timeout -k 10 200 python bench.py --steps 60 --warmup 20 2>/dev/null | tail -1 > gpurun_out/ap_bench_on_$i.json
done
python3 - <<'PY'
import json
for k in ("off_1", "on_1", "off_2", "on_2"):
    try:
        d = json.load(open("gpurun_out/ap_bench_%s.json" % k))
        print(k, d["value"], d["ms_per_step"], d.get("clock_ghz_under_load"), d["roofline"]["frac"], d.get("fallback_launches"))
    except Exception as e:
        print(k, "failed", e)
PY
