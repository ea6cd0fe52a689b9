#!/bin/bash
# PMC counters of the fp16 convolution family on tools/conv_f16_bench.py (separate passes with --kernel-trace only, as gpurun requires):
# matrix-pipe busy fraction, wave-cycle split, LDS bank conflicts, HBM bytes (FETCH_SIZE x 2, WRITE_SIZE).  usage: tools/gpu_pmc_f16.sh <tag>
TAG=${1:-r03_f16_pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o t -- python tools/conv_f16_bench.py > $OUT/p$i.log 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); ms = collections.defaultdict(float)
for f in glob.glob('$OUT/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
        cnt[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[k] += 1
for r in csv.DictReader(open(glob.glob('$OUT/p1/*kernel_trace.csv')[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    ms[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6
lines = ['# fp16 kernels on tools/conv_f16_bench.py (batch 8; 13 launches per shape incl. warm-up): rocprofv3 --pmc passes of tools/gpu_pmc_f16.sh',
         '# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128); wave-cycle split wait_any (s_waitcnt / barrier) | wait_inst (issue stall) | active;',
         '# bank_conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; HBM bytes per launch = FETCH_SIZE KiB x 2 (gfx950 correction) / WRITE_SIZE KiB',
         f'{"kernel":44s} {"launches":>8s} {"ms":>8s} {"mfma_busy":>9s} {"wait_any":>8s} {"wait_inst":>9s} {"active":>6s} {"bank_conf":>9s} {"read MB":>8s} {"write MB":>8s}']
for k, c in sorted(cnt.items(), key=lambda kv: -ms[kv[0]]):
    if not k.startswith('f16::') or ms[k] <= 0: continue
    gui, wc, nn = c['GRBM_GUI_ACTIVE'] or 1, c.get('SQ_WAVE_CYCLES', 0) or 1, max(n[k], 1)
    lines.append(f'{k:44s} {nn:8d} {ms[k]:8.2f} {c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / gui / 128:9.3f} {c.get("SQ_WAIT_ANY", 0) / wc:8.3f} '
                 f'{c.get("SQ_WAIT_INST_ANY", 0) / wc:9.3f} {c.get("SQ_ACTIVE_INST_ANY", 0) / wc:6.3f} {c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1):9.3f} '
                 f'{c.get("FETCH_SIZE", 0) * 2048 / nn / 1e6:8.1f} {c.get("WRITE_SIZE", 0) * 1024 / nn / 1e6:8.1f}')
open('gpurun_out/${TAG}_summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*.db' -delete
