#!/bin/bash
# Round evidence in one call, all on the SAME build and box: kernel stats + per-layer table + bench
# lines of every workload, MFMA / issue counters per layer class, per-layer and whole-step HBM traffic, FETCH_SIZE
# calibration.  Everything lands in gpurun_out/ (copy the summaries to profiles/).   usage: TAG=r03 bash scripts/gpu_evidence.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-r06}
python3 -c "import sys; sys.path.insert(0, '.'); from pointcloududa_amd._lib import csrc_hash; print('csrc_sha256', csrc_hash())" | tee gpurun_out/${TAG}_build_hash.txt
# (the whole-step traffic first: bench.py quotes roofline.traffic from profiles/rNN_pmc_traffic.csv of THIS build's hash)
TAG=$TAG bash scripts/gpu_pmc_step.sh > gpurun_out/${TAG}_pmc_step.log 2>&1; rc=$?; echo "pmc_step rc=$rc"
# (only a pass that succeeded AND produced rows replaces the committed summary)
if [ $rc -eq 0 ] && [ "$(grep -vc '^#' gpurun_out/${TAG}_pmc_traffic.csv 2>/dev/null)" -gt 1 ]; then cp gpurun_out/${TAG}_pmc_traffic.csv profiles/${TAG}_pmc_traffic.csv; else echo "pmc_traffic NOT copied"; fi
TAG=$TAG bash scripts/gpu_profile.sh > gpurun_out/${TAG}_profile.log 2>&1; echo "profile rc=$?"
for wl in unet_d2 mmwhs_uda uda_512 mscmrseg_224; do
  python3 bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$wl.json; echo "$wl: $(cut -c1-160 gpurun_out/${TAG}_bench_$wl.json)"
done
python3 bench.py --precision bf16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_full_uda_bf16.json; echo "bf16: $(cut -c1-160 gpurun_out/${TAG}_bench_full_uda_bf16.json)"
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_line.json      # (the default command's line, with cpu_baseline and the traffic of THIS build)
# the N > 1 code path on one GPU: RCCL all-reduces in a one-rank group (eager, then replayed from two hipGraphs), and the
# control flow of --gpus 2 with two ranks sharing the GPU over gloo (a rehearsal, not a measurement)
PCUDA_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_forced_collectives_1rank.json
PCUDA_GRAPH=1 PCUDA_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_forced_collectives_1rank_graph.json
PCUDA_SHARE_GPU=1 PCUDA_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_rehearsal_2ranks_one_gpu_gloo.json
# per-layer table of the 224x224 workload (the table that exposed round 4's plan rules)
PCUDA_DSTREAMS=0 PCUDA_PROF_DUMP=gpurun_out/${TAG}_layers224.csv python3 bench.py --workload mscmrseg_224 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 scripts/layer_table.py gpurun_out/${TAG}_layers224.csv 60 > gpurun_out/${TAG}_layer_table_224.txt
PCUDA_TIMELINE=1 python3 scripts/step_timeline.py > gpurun_out/${TAG}_step_timeline.txt 2>&1
# (g6432 / g3264: the register-window weight gradient wgrad3r_kernel; p96 / p192: the 1x1 NT-GEMM wgrad1_kernel; d1: d1_fwd_kernel -- round-4 review item 5)
CASES="g32 g6432 g3264 g64 g128 g256 b512 p96 p192 d1 d2 d4" TAG=$TAG NSETS=3 bash scripts/pmc_conv.sh > gpurun_out/${TAG}_pmc_conv.log 2>&1; echo "pmc_conv rc=$?"
python3 scripts/mfma_table.py gpurun_out/${TAG}_pmc_conv.csv gpurun_out/${TAG}_mfma_counters.csv > /dev/null; echo "mfma table rc=$?"
TAG=$TAG bash scripts/pmc_layers.sh > gpurun_out/${TAG}_pmc_layers.log 2>&1; echo "pmc_layers rc=$?"
TAG=$TAG bash scripts/micro/fetch_calib.sh > gpurun_out/${TAG}_fetch_calib.log 2>&1; echo "fetch_calib rc=$?"
# per-phase stamps (make CLK=1 -> lib/libpcuda_clk.so) incl. the eight-wave kernel, and the switched-off-phase bounds (XFLAGS=-DPCUDA_WEXP)
if [ -f pointcloududa_amd/lib/libpcuda_clk.so ]; then PCUDA_LIB=pointcloududa_amd/lib/libpcuda_clk.so python3 scripts/clk_micro.py g32 g64 g128 g256 d2 d4 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_clk_phases.txt; echo "clk rc=$?"; fi
if [ -f pointcloududa_amd/lib/libpcuda_wexp.so ]; then CASES="g32 g6432 g64 g128 g256 b512 d2 d3 d4" DBGS="0 24 32 64 56 120" bash scripts/micro/wexp.sh > gpurun_out/${TAG}_wexp_bounds.txt 2>&1; echo "wexp rc=$?"; fi
ls gpurun_out/${TAG}_*
