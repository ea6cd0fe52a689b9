#!/bin/bash
# PMC counters of the bench's kernels (counter passes with --kernel-trace only, as gpurun requires).
# usage: tools/gpu_pmc_bench.sh <tag> ; writes gpurun_out/<tag>/pmc_summary.txt (per kernel, summed over its launches)
TAG=${1:-r02pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o t -- python bench.py --steps 2 --warmup 1 --pipeline-depth 1 --profile-steps 0 --no-cpu-baseline --no-second-config --no-train-step --no-eval-loop --graph off > $OUT/p$i.log 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('$OUT/p*/*counter_collection.csv'):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and f.endswith('p1/t_counter_collection.csv'.split('/')[-1]) and '/p1/' in f: n[k] += 1
lines = []
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
    gui = d.get('GRBM_GUI_ACTIVE', 0) / 2.0 if 'SQ_ACTIVE_INST_VALU' in d and 'SQ_WAVE_CYCLES' in d else d.get('GRBM_GUI_ACTIVE', 0)
    if gui < 1e5: continue
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    lines.append(f"{k:48s} launches={n[k]:4d} gui_cycles={gui:.3e} mfma_busy={d.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/gui/128:.3f} "
                 f"mops_f32={d.get('SQ_INSTS_VALU_MFMA_MOPS_F32',0):.3e} wait_any/wave={d.get('SQ_WAIT_ANY',0)/wc:.3f} wait_inst/wave={d.get('SQ_WAIT_INST_ANY',0)/wc:.3f} "
                 f"active/wave={d.get('SQ_ACTIVE_INST_ANY',0)/wc:.3f} valu/wave={d.get('SQ_ACTIVE_INST_VALU',0)/wc:.3f} lds/wave={d.get('SQ_ACTIVE_INST_LDS',0)/wc:.3f} "
                 f"wait_lds/wave={d.get('SQ_WAIT_INST_LDS',0)/wc:.3f} bank_conf/lds_active={d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_LDS_IDX_ACTIVE',0),1):.3f} "
                 f"lds_idx_active/gui/256={d.get('SQ_LDS_IDX_ACTIVE',0)/gui/256:.3f} busy_cu/gui={d.get('SQ_BUSY_CU_CYCLES',0)/gui:.1f}")
open('$OUT/pmc_summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:14]))
PY
