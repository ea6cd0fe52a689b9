#!/bin/bash
# Kernel-trace durations of conv_f16_kernel under environment switches.  usage: tools/f16_env.sh "NAME=1" "OTHER=1" ...  ("-" = no switch)
OUT=$GRAFT_REPO_ROOT/gpurun_out/f16_env
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for V in "$@"; do
  i=$((i+1)); E=$V; [ "$V" = - ] && E=SHG_NOTHING=1
  ( cd $GRAFT_REPO_ROOT && env $E SHG_F16_FWD_ONLY=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r$i -o t -- python tools/conv_f16_bench.py > $OUT/r$i.log 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - "$@" <<PY
import csv, glob, collections, sys
for i, v in enumerate(sys.argv[1:]):
    d = collections.defaultdict(list)
    for f in glob.glob('$OUT/r%d/*kernel_trace.csv' % (i + 1)):
        for r in csv.DictReader(open(f)):
            if 'conv_f16_kernel' not in r['Kernel_Name']: continue
            d[r['Kernel_Name'].split('(')[0].replace('void f16::', '')].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3)
    print('== %s' % v)
    for k, t in d.items():          # launches of one kernel come in runs of 14 per shape, in bench order
        runs = [sorted(t[j:j + 14]) for j in range(0, len(t), 14)]
        print('   %-36s' % k, ' '.join('%7.1f' % r[len(r) // 2] for r in runs))
PY
