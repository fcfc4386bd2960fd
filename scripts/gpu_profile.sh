#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats, per-layer table (in-library HIP events) and the bench line.  Everything lands in gpurun_out/ (copy the summaries to profiles/).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
TAG=${TAG:-r03}
# (kernel durations with the discriminator streams serialised, as in bench.py's roofline pass: concurrent kernels are each billed the shared time)
PCUDA_DSTREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_bench_$TAG.log 2>&1
tail -1 gpurun_out/prof_bench_$TAG.log | cut -c1-300
F=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" gpurun_out/${TAG}_kernel_stats.csv && head -12 "$F"
# (HBM traffic counters: scripts/gpu_pmc.sh, on single layers -- a --pmc pass over the whole step serialises
# thousands of dispatches and takes tens of minutes)
PCUDA_DSTREAMS=0 PCUDA_PROF_DUMP=gpurun_out/${TAG}_layers.csv python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_prof.log 2>&1
python3 scripts/layer_table.py gpurun_out/${TAG}_layers.csv 60 > gpurun_out/${TAG}_layer_table.txt
tail -3 gpurun_out/${TAG}_layer_table.txt
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_line.json
cut -c1-200 gpurun_out/${TAG}_bench_line.json
