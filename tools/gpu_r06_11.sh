#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_config5.py tests/test_gpu_backward.py tests/test_gpu_fp16.py tests/test_gpu_train_graph.py -x -q > gpurun_out/r06r_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r06r_pytest.log
tail -3 gpurun_out/r06r_pytest.log
for V in 0 1; do for F in "" "--fp16"; do echo "== SHG_LINEAR_GAIN_ON_WEIGHTS=$V $F"; SHG_LINEAR_GAIN_ON_WEIGHTS=$V python tools/train_step_bench.py --steps 4 $F 2>/dev/null | grep -E "^G phase|bias_act"; done; done
