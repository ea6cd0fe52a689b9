#!/bin/bash
# PMC counters of the F(4x4) kernel on the timing layers of tools/wino4_check.py (counter passes with --kernel-trace only).
OUT=$GRAFT_REPO_ROOT/gpurun_out/w4pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  ( cd $GRAFT_REPO_ROOT && SHG_VARIANT=${1:-} timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o t -- python tools/wino4_check.py > $OUT/p$i.log 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$OUT/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
        if 'wino' not in k or 'weight' in k: continue
        agg[k + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in sorted(agg.items()):
    print(k)
    print('   ', ' '.join(f'{c}={v:.4g}' for c, v in sorted(d.items())))
PY
