#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_wgrad_wino.py tests/test_gpu_fp16.py tests/test_gpu_fp16_train.py tests/test_gpu_config5.py -x -q > gpurun_out/r06d_pytest_focus.log 2>&1; echo "rc $?" >> gpurun_out/r06d_pytest_focus.log
tail -5 gpurun_out/r06d_pytest_focus.log
python tools/wgrad_bench.py 2>/dev/null | grep -v amdgpu > gpurun_out/r06d_wgrad_bench.txt; cat gpurun_out/r06d_wgrad_bench.txt
python tools/conv_f16_bench.py 2>/dev/null | grep "^conv" > gpurun_out/r06d_conv_f16_bench.txt; cat gpurun_out/r06d_conv_f16_bench.txt
