#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "shu or generator or heterogeneous or config5" > gpurun_out/r06n_pytest_shu.log 2>&1; echo "rc $?" >> gpurun_out/r06n_pytest_shu.log
tail -3 gpurun_out/r06n_pytest_shu.log
timeout 300 python tools/shu_floor.py 2>&1 | grep -v amdgpu > gpurun_out/r06n_shu_floor.txt; cat gpurun_out/r06n_shu_floor.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-config --no-train-step --no-eval-loop 2>/dev/null | grep "^{" > gpurun_out/r06n_bench.json; python - <<'P'
import json
d=json.loads(open('gpurun_out/r06n_bench.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d['shu'])[:900])
P
for D in 2 4; do python bench.py --steps 20 --warmup 5 --pipeline-depth $D --no-cpu-baseline --no-second-config --no-train-step --no-eval-loop --profile-steps 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('depth', $D, d['value'], d['ms_per_step'])"; done
