#!/bin/bash
# PMC counters for the conv microbench (separate pass, kernel-trace only as gpurun requires).
# usage: tools/gpu_pmc.sh <tag> "<counters>" [conv_bench filter]
TAG=$1; CNT=$2; FLT=$3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -o pmc -- python ${PMC_SCRIPT:-tools/conv_bench.py} "$FLT" > $OUT/stdout.log 2>&1
echo "exit $?" >> $OUT/stdout.log
tail -5 $OUT/stdout.log
ls $OUT
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/*counter_collection.csv')
if f:
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    for k, d in agg.items():
        if 'conv_mfma' in k or 'fir' in k:
            print(k); print('   ', {c: f'{v:.4g}' for c, v in d.items()})
PY
