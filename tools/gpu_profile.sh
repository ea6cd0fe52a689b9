#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench; summaries land in gpurun_out/prof_<tag>/
TAG=${1:-r01}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $OUT/bench_stdout.log 2>&1
echo "rocprof exit $?" >> $OUT/bench_stdout.log
find $OUT -name "*stats*" | head
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -40 "$F"
# keep the merged-back payload small: drop the raw per-dispatch trace if it is huge
find $OUT -name "*kernel_trace.csv" -size +20M -delete
tail -3 $OUT/bench_stdout.log
