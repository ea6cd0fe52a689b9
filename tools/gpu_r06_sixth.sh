#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv1x1 or mfma_conv" > gpurun_out/r06f_pytest_focus.log 2>&1; echo "rc $?" >> gpurun_out/r06f_pytest_focus.log
tail -3 gpurun_out/r06f_pytest_focus.log
python tools/conv1x1_bench.py 2>/dev/null | grep "^1x1" > gpurun_out/r06f_conv1x1_bench.txt; cat gpurun_out/r06f_conv1x1_bench.txt
