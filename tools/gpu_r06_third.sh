#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_graph.py tests/test_gpu_wgrad_wino.py -x -q > gpurun_out/r06c_pytest_focus.log 2>&1; echo "rc $?" >> gpurun_out/r06c_pytest_focus.log
tail -5 gpurun_out/r06c_pytest_focus.log
python tools/wgrad_bench.py 2>/dev/null | grep -v amdgpu > gpurun_out/r06c_wgrad_bench.txt; cat gpurun_out/r06c_wgrad_bench.txt
python tools/wgrad_bench.py --batch 16 2>/dev/null | grep -v amdgpu > gpurun_out/r06c_wgrad_bench_b16.txt; cat gpurun_out/r06c_wgrad_bench_b16.txt
