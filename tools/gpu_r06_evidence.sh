#!/bin/bash
# Round 6 evidence set in one visit: profile set (kernel stats of the bench command single-stream / pipelined, bench lines 512x16 and 256x32,
# training-step kernel stats fp32 / fp16, per-shape tables), HBM traffic per kernel class and per conv_wino4 instantiation (PMC), MFMA
# utilisation of the forward path (PMC), counters of the weight-gradient kernels, the SHU floor probe.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_round_profile.sh r06 > gpurun_out/r06_round_profile.log 2>&1
bash tools/gpu_traffic.sh r06 > gpurun_out/r06_traffic.log 2>&1
tail -12 gpurun_out/r06_traffic.log
bash tools/gpu_pmc_bench.sh r06pmc > gpurun_out/r06_pmc_bench.log 2>&1
tail -12 gpurun_out/r06_pmc_bench.log
bash tools/gpu_pmc_cmd.sh r06_wgrad_pmc wgrad python tools/wgrad_bench.py > gpurun_out/r06_wgrad_pmc.log 2>&1
tail -12 gpurun_out/r06_wgrad_pmc.log
bash tools/gpu_pmc_cmd.sh r06_conv1x1_pmc conv python tools/conv1x1_bench.py > gpurun_out/r06_conv1x1_pmc.log 2>&1
tail -6 gpurun_out/r06_conv1x1_pmc.log
timeout 300 python tools/shu_floor.py > gpurun_out/r06_shu_floor.txt 2>&1
python tools/conv1x1_bench.py 2>/dev/null | grep "^1x1" > gpurun_out/r06/r06_conv1x1_bench.txt
rm -rf gpurun_out/pmc_traffic_* gpurun_out/r06pmc/p1 gpurun_out/r06pmc/p2
tail -30 gpurun_out/r06_round_profile.log
