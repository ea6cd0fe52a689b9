#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (separate passes: FETCH_SIZE, WRITE_SIZE; kernel-trace only).
# Writes gpurun_out/traffic_<tag>.json: per kernel class bytes per launch (FETCH_SIZE doubled per MI355X_MICROARCH.md).
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_$C
  rm -rf $OUT; mkdir -p $OUT
  ( cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o t -- python bench.py --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-second-config --no-train-step --no-eval-loop --graph off > $OUT/stdout.log 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: dict(launches=0, FETCH_SIZE=0.0, WRITE_SIZE=0.0))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_traffic_%s/*counter_collection.csv' % c)
    if not f: continue
    seen = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != c: continue
        full = r['Kernel_Name'].split('(')[0].replace('void ', '')
        k = full.split('<')[0]
        res[k][c] += float(r['Counter_Value'])
        seen[k] += 1
        if k == 'conv_wino4_kernel' and '<' in full:      # per instantiation <TY, TX>: the thin / medium / thick layers separately
            ki = full.replace(' ', '')
            res[ki][c] += float(r['Counter_Value'])
            seen[ki] += 1
    for k, n in seen.items(): res[k]['launches'] = max(res[k]['launches'], n)
out = {}
for k, d in res.items():
    n = max(d['launches'], 1)
    # counters are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section)
    out[k] = dict(launches=n, read_bytes_per_launch=d['FETCH_SIZE'] * 1024 * 2 / n, write_bytes_per_launch=d['WRITE_SIZE'] * 1024 / n,
                  fetch_size_kib_raw=d['FETCH_SIZE'], write_size_kib_raw=d['WRITE_SIZE'])
json.dump(out, open('gpurun_out/traffic_$TAG.json', 'w'), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -kv[1]['read_bytes_per_launch'] * kv[1]['launches'])[:10]:
    print(f"{k:32s} launches={d['launches']:4d} read/launch={d['read_bytes_per_launch']/1e6:9.1f} MB write/launch={d['write_bytes_per_launch']/1e6:9.1f} MB")
PY
