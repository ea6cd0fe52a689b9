#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/train_step_bench.py --steps 3 2>&1 | grep -v amdgpu > gpurun_out/r06g_train_fp32.txt; head -40 gpurun_out/r06g_train_fp32.txt
python tools/train_step_bench.py --steps 3 --fp16 2>&1 | grep -v amdgpu > gpurun_out/r06g_train_fp16.txt; head -50 gpurun_out/r06g_train_fp16.txt
python tools/conv1x1_bench.py 2>/dev/null | grep "^1x1" > gpurun_out/r06g_conv1x1_bench.txt
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r06g_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r06g_pytest.log
tail -4 gpurun_out/r06g_pytest.log
