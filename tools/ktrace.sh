#!/bin/bash
# per-kernel (name, grid) duration table of a command under rocprofv3 --kernel-trace.  usage: tools/ktrace.sh <tag> <filter-regex> -- <cmd...>
TAG=$1; FLT=$2; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- "$@" > $OUT/stdout.log 2>&1 )
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections, re
f = glob.glob('$OUT/*kernel_trace.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:64]
    agg[(k, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1) * int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if re.search(r'$FLT', k):
        v = sorted(v); print(f'{k:64s} wgs={g:7d} n={len(v):4d} min={v[0]:9.1f} med={v[len(v)//2]:9.1f} sum={sum(v)/1e3:9.2f} ms')
PY
