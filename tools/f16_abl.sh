#!/bin/bash
# Kernel-trace durations of conv_f16_kernel per launch geometry for a list of study variants (tools/_variants/libshgan_hip_<tag>.so, built with
# `python sh-gan_amd/build.py --variant=<tag> -DSHG_F16_ABL=<bits> ...`).  Event timing of tools/conv_f16_bench.py includes the weight packing
# launches and the host; this does not.  usage: tools/f16_abl.sh "<tag> <tag> ..."   ("base" = the product library)
OUT=$GRAFT_REPO_ROOT/gpurun_out/f16_abl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in $1; do
  VV=$V; [ "$V" = base ] && VV=
  ( cd $GRAFT_REPO_ROOT && SHG_VARIANT=$VV SHG_F16_FWD_ONLY=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$V -o t -- python tools/conv_f16_bench.py > $OUT/$V.log 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for v in "$1".split():
    d = collections.defaultdict(list)
    for f in glob.glob('$OUT/%s/*kernel_trace.csv' % v):
        for r in csv.DictReader(open(f)):
            if 'conv_f16_kernel' not in r['Kernel_Name']: continue
            k = r['Kernel_Name'].split('(')[0].replace('void f16::', '') + ' grid %sx%s' % (int(r['Grid_Size_X']) // 256, r['Grid_Size_Y'])
            d[k].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3)
    print('== %s' % v)
    for k, t in d.items():
        t = sorted(t)
        print('   %-48s n=%3d  median %8.1f us' % (k, len(t), t[len(t) // 2]))
PY
