#!/bin/bash
# Everything a profiles/<tag>_* set holds, in one GPU-box visit: rocprofv3 kernel stats of the bench command, the bench
# JSON lines (512x16 with the CPU baseline, 256x32), the per-layer micro-benchmarks.  Output: gpurun_out/<tag>/.
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (bench runs warmup + steps + 3 single-stream latency steps = 10 forward passes here: the divisor of prof_summary.py; --graph off: the
#  auto mode would add its trial steps and the capture warm-up)
# kernel durations: one batch at a time on one stream (what bench.py's roofline block times with HIP events), then the same command
# with the default three-stream pipeline (kernels of different batches overlap: longer individual durations, shorter wall time)
for MODE in single pipelined; do
  D=1; [ $MODE = pipelined ] && D=3
  ( cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 5 --warmup 2 --pipeline-depth $D --profile-steps 0 --graph off --no-cpu-baseline --no-second-config --no-train-step --no-eval-loop > $OUT/prof_stdout_$MODE.log 2>&1 )
  cd $GRAFT_REPO_ROOT
  F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  SUF=""; [ $MODE = pipelined ] && SUF="_pipelined"
  cp "$F" $OUT/${TAG}_bench512x16${SUF}_kernel_stats.csv
  python tools/prof_summary.py "$F" 10 > $OUT/${TAG}_summary${SUF}.txt
  rm -rf $OUT/prof
  cd /tmp
done
cd $GRAFT_REPO_ROOT
python bench.py --steps 40 --warmup 6 ${BENCH_ARGS} > $OUT/${TAG}_bench512x16.json.log 2> $OUT/bench512.err
python bench.py --steps 40 --warmup 6 --resolution 256 --no-cpu-baseline > $OUT/${TAG}_bench256x32.json.log 2> $OUT/bench256.err
# BASELINE config 5: kernel stats of the training step, float32 and with fp16 blocks (tools/train_step_bench.py: 1 warm-up + 2 timed +
# 1 instrumented step = 4 steps: the divisor), and the per-shape table of the fp16 convolution family
for MODE in fp32 fp16; do
  FL=""; [ $MODE = fp16 ] && FL="--fp16"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tprof -o step -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --steps 2 $FL > $OUT/${TAG}_train_${MODE}_stdout.log 2>&1 )
  F=$(find $OUT/tprof -name "*kernel_stats.csv" | head -1)
  cp "$F" $OUT/${TAG}_train512x8_${MODE}_kernel_stats.csv
  python tools/prof_summary.py "$F" 4 > $OUT/${TAG}_train_${MODE}_summary.txt
  rm -rf $OUT/tprof
done
python tools/conv_f16_bench.py > $OUT/${TAG}_conv_f16_bench.txt 2>/dev/null
python tools/conv_bench.py > $OUT/${TAG}_conv_bench.txt 2>/dev/null
python tools/conv_bench_down.py > $OUT/${TAG}_conv_bench_down.txt 2>/dev/null
python tools/fir_bench.py > $OUT/${TAG}_fir_bench.txt 2>/dev/null
python tools/wgrad_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_wgrad_bench.txt
python tools/pointwise_bench.py 2>/dev/null | grep '^\[' > $OUT/${TAG}_pointwise_bench.txt
cat $OUT/${TAG}_summary.txt; head -c 300 $OUT/${TAG}_bench512x16.json.log; echo; head -c 300 $OUT/${TAG}_bench256x32.json.log
