"""Attribute the HBM read traffic of conv_wino4's three instantiations (profiles/traffic_512x16.json, tools/gpu_traffic.sh: PMC FETCH_SIZE x 2 /
WRITE_SIZE per launch) to its two known causes: the window halo of a tile ((4 TY + 2)(4 TX + 8) floats fetched per 16 TY TX outputs) and the
re-read of the input by every 64-output-channel tile (O / 64 workgroups per pixel tile).  The headline workload: FFHQ-512 generator, batch 16 --
per resolution one encoder conv0 and one synthesis conv1 of C(res) -> C(res) channels.

  python tools/traffic_attribution.py [profiles/traffic_512x16.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'traffic_512x16.json')
d = json.load(open(path))
N = 16
CH = {512: 64, 256: 128, 128: 256, 64: 512, 32: 512}
INST = {512: (1, 32), 256: (1, 32), 128: (2, 16), 64: (4, 8), 32: (4, 8)}
rows = {}
for res, c in CH.items():
    ty, tx = INST[res]
    alg = N * c * res * res * 4 / 1e9
    halo = (4 * ty + 2) * (4 * tx + 8) / (16.0 * ty * tx)
    ot = c // 64
    r = rows.setdefault((ty, tx), dict(launches=0, alg_r=0.0, halo_r=0.0, full_r=0.0, alg_w=0.0))
    r['launches'] += 2
    r['alg_r'] += 2 * alg
    r['halo_r'] += 2 * alg * halo
    r['full_r'] += 2 * alg * halo * ot
    r['alg_w'] += 2 * alg
print(f'{"instantiation":22s} {"launches":>8s} | per launch, GB: {"algorithmic":>11s} {"x halo":>8s} {"x halo x O/64":>14s} {"measured read":>14s} {"measured write":>15s} {"algorithmic write":>18s}')
for (ty, tx), r in rows.items():
    key = f'conv_wino4_kernel<{ty},{tx}>'
    m = d.get(key)
    n = r['launches']
    mr = m['read_bytes_per_launch'] / 1e9 if m else float('nan')
    mw = m['write_bytes_per_launch'] / 1e9 if m else float('nan')
    print(f'{key:22s} {n:8d} | {"":16s} {r["alg_r"] / n:11.3f} {r["halo_r"] / n:8.3f} {r["full_r"] / n:14.3f} {mr:14.3f} {mw:15.3f} {r["alg_w"] / n:18.3f}')
    if m:
        print(f'{"":22s} read = {mr / (r["alg_r"] / n):.2f} x algorithmic: halo {r["halo_r"] / r["alg_r"]:.2f} x, the O/64 re-reads would make it '
              f'{r["full_r"] / r["alg_r"]:.2f} x if none hit the L2 / Infinity Cache -> {100 * (1 - (mr - r["halo_r"] / n) / max(r["full_r"] / n - r["halo_r"] / n, 1e-9)):.0f} % of the re-reads served on chip')
