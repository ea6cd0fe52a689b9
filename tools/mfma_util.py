"""MFMA utilisation table from a tools/gpu_pmc_bench.sh run: python tools/mfma_util.py gpurun_out/<tag> > profiles/<tag>_mfma_util.txt"""
import collections, csv, glob, sys
d = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(glob.glob(d + '/p1/*counter_collection.csv')[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:52]
    cnt[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        n[k] += 1
ms = collections.defaultdict(float)
for r in csv.DictReader(open(glob.glob(d + '/p1/*kernel_trace.csv')[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:52]
    ms[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6
print('# MFMA utilisation per kernel of `python bench.py --steps 2 --warmup 1` (512x512, batch 16), rocprofv3 --pmc pass of tools/gpu_pmc_bench.sh')
print('# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)   [GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs]')
print('# executed TFLOP/s = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 flop / kernel time (profiled pass: clocks run ~3 % lower than unprofiled)')
print('# eff_clock = GRBM_GUI_ACTIVE / 8 / kernel time; wave-cycle split: wait_any (s_waitcnt / barrier), wait_inst (issue stall), active')
print(f'{"kernel":52s} {"launches":>8s} {"ms":>8s} {"mfma_busy":>9s} {"exec TF/s":>9s} {"frac157.3":>9s} {"clk GHz":>7s} {"wait_any":>8s} {"wait_inst":>9s} {"active":>6s}')
for k, c in sorted(cnt.items(), key=lambda kv: -ms[kv[0]]):
    if c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) <= 0 or ms[k] <= 0:
        continue
    gui, wc = c['GRBM_GUI_ACTIVE'], c.get('SQ_WAVE_CYCLES', 0) or 1
    tf = c['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512 / (ms[k] * 1e-3) / 1e12
    print(f'{k:52s} {n[k]:8d} {ms[k]:8.2f} {c["SQ_VALU_MFMA_BUSY_CYCLES"] / gui / 128:9.3f} {tf:9.1f} {tf / 157.3:9.3f} {gui / 8 / (ms[k] * 1e-3) / 1e9:7.2f} '
          f'{c.get("SQ_WAIT_ANY", 0) / wc:8.3f} {c.get("SQ_WAIT_INST_ANY", 0) / wc:9.3f} {c.get("SQ_ACTIVE_INST_ANY", 0) / wc:6.3f}')
