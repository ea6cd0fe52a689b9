#!/bin/bash
# PMC counters of the kernels of any command (separate passes with --kernel-trace only, as gpurun requires): matrix-pipe busy fraction, wave-cycle
# split, LDS bank conflicts, HBM bytes (FETCH_SIZE x 2, WRITE_SIZE).  usage: tools/gpu_pmc_cmd.sh <tag> <kernel-name substring> <command...>
TAG=$1; FILT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o t -- "$@" > $OUT/p$i.log 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); ms = collections.defaultdict(float)
for f in glob.glob('$OUT/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:52]
        cnt[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[k] += 1
for r in csv.DictReader(open(glob.glob('$OUT/p1/*kernel_trace.csv')[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:52]
    ms[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6
lines = ['# rocprofv3 --pmc passes of tools/gpu_pmc_cmd.sh: $*',
         '# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128 [x 4 SIMDs x 32 CU-groups as in tools/gpu_pmc_f16.sh]); wave-cycle split wait_any | wait_inst | active;',
         '# valu / lds / vmem = SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES; bank_conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; HBM MB per launch = FETCH_SIZE x 2 / WRITE_SIZE',
         f'{"kernel":52s} {"n":>4s} {"ms":>8s} {"mfma":>6s} {"w_any":>6s} {"w_inst":>6s} {"active":>6s} {"valu":>6s} {"lds":>6s} {"vmem":>6s} {"w_lds":>6s} {"bankc":>6s} {"rd MB":>8s} {"wr MB":>8s}']
for k, c in sorted(cnt.items(), key=lambda kv: -ms[kv[0]]):
    if '$FILT' not in k or ms[k] <= 0: continue
    gui, wc, nn = c['GRBM_GUI_ACTIVE'] or 1, c.get('SQ_WAVE_CYCLES', 0) or 1, max(n[k], 1)
    wc2 = wc / 2 if c.get('SQ_ACTIVE_INST_VALU') is not None and 'SQ_WAIT_ANY' in c else wc     # SQ_WAVE_CYCLES is collected in two passes
    f = lambda name, d=wc2: c.get(name, 0) / d
    lines.append(f'{k:52s} {nn:4d} {ms[k]:8.2f} {c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / gui / 128:6.3f} {f("SQ_WAIT_ANY"):6.3f} {f("SQ_WAIT_INST_ANY"):6.3f} {f("SQ_ACTIVE_INST_ANY"):6.3f} '
                 f'{f("SQ_ACTIVE_INST_VALU"):6.3f} {f("SQ_ACTIVE_INST_LDS"):6.3f} {f("SQ_ACTIVE_INST_VMEM"):6.3f} {f("SQ_WAIT_INST_LDS"):6.3f} '
                 f'{c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1):6.3f} {c.get("FETCH_SIZE", 0) * 2048 / nn / 1e6:8.1f} {c.get("WRITE_SIZE", 0) * 1024 / nn / 1e6:8.1f}')
open('gpurun_out/${TAG}_summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*.db' -delete
