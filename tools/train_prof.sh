cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trainprof
rm -rf $OUT; mkdir -p $OUT
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python tools/train_step_bench.py --steps 3 > $OUT/stdout.log 2>&1 )
cd $GRAFT_REPO_ROOT
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py "$F" 4 | head -45
tail -12 $OUT/stdout.log
rm -rf $OUT/prof
