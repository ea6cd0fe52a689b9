#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_wgrad_wino.py -x -q > gpurun_out/r06k_pytest_focus.log 2>&1; echo "rc $?" >> gpurun_out/r06k_pytest_focus.log
tail -3 gpurun_out/r06k_pytest_focus.log
python tools/launch_outliers.py --ratio 0.5 --min-us 60 2>/dev/null | grep -A8 "^conv_wgrad  " | head -12
timeout 420 python bench.py --gpus 2 --all-blocks --train-steps 1 --steps 2 --warmup 1 --no-cpu-baseline --no-second-config --no-eval-loop --watchdog 120 > gpurun_out/r06j_bench_2ranks.log 2>&1; echo rc $?
grep -n "File \|^Thread\|Current thread" gpurun_out/r06j_bench_2ranks.log | tail -60
