#!/bin/bash
# Round 6, second visit: the unmasked weight-gradient form (tests + per-shape table), the split-graph world-2 test, the rest of the suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_graph.py tests/test_gpu_fp16.py -x -q > gpurun_out/r06b_pytest_focus.log 2>&1; echo "rc $?" >> gpurun_out/r06b_pytest_focus.log
tail -15 gpurun_out/r06b_pytest_focus.log
python tools/wgrad_bench.py 2>/dev/null | grep -v amdgpu > gpurun_out/r06b_wgrad_bench.txt; cat gpurun_out/r06b_wgrad_bench.txt
python tools/wgrad_bench.py --batch 16 2>/dev/null | grep -v amdgpu > gpurun_out/r06b_wgrad_bench_b16.txt; cat gpurun_out/r06b_wgrad_bench_b16.txt
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_backward.py --deselect tests/test_gpu_train_graph.py --deselect tests/test_gpu_fp16.py > gpurun_out/r06b_pytest_rest.log 2>&1; echo "rc $?" >> gpurun_out/r06b_pytest_rest.log
tail -8 gpurun_out/r06b_pytest_rest.log
