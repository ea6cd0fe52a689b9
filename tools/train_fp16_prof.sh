#!/bin/bash
# fp16 / fp32 training step (config 5): rocprofv3 kernel stats.  usage (GPU box): bash tools/train_fp16_prof.sh [tag] [--fp16]
TAG=${1:-r3_trainprof16}; shift
cd "$(dirname "$0")/.." && R=$PWD
OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o step -- python $R/tools/train_step_bench.py --steps 2 "$@" > $OUT/stdout.log 2>&1
cd $R
F=$(find $OUT -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && python tools/prof_summary.py $F 4 > gpurun_out/${TAG}_summary.txt 2>&1
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
