#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_backward.py tests/test_gpu_wgrad_wino.py tests/test_gpu_fp16.py tests/test_gpu_config5.py -x -q > gpurun_out/r06e_pytest_focus.log 2>&1; echo "rc $?" >> gpurun_out/r06e_pytest_focus.log
tail -5 gpurun_out/r06e_pytest_focus.log
python tools/conv1x1_bench.py 2>/dev/null | grep "^1x1" > gpurun_out/r06e_conv1x1_bench.txt; cat gpurun_out/r06e_conv1x1_bench.txt
python tools/train_step_bench.py --steps 3 2>&1 | grep -v amdgpu | tail -5
