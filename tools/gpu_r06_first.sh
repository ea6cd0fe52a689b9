#!/bin/bash
# Round 6, first GPU visit: the parity suite, then the round's profile set, the SHU floor probe and the counter passes.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_pytest.log
tail -3 gpurun_out/r06_pytest.log
bash tools/gpu_round_profile.sh r06 > gpurun_out/r06_round_profile.log 2>&1
tail -30 gpurun_out/r06_round_profile.log
timeout 300 python tools/shu_floor.py > gpurun_out/r06_shu_floor.txt 2>&1
tail -30 gpurun_out/r06_shu_floor.txt
