"""Throughput of the on-device freeform-mask path: host record generation (numpy RNG in the reference's order) + H2D +
HIP rasteriser, vs the reference-style host path (Pillow).  usage: python tools/mask_bench.py [s] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import shgan_amd
from shgan_amd import data, masks
s = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
np.random.seed(0); masks.random_masks(8, s, device='cuda'); torch.cuda.synchronize()
np.random.seed(1); t0 = time.perf_counter(); m = masks.random_masks(n, s, device='cuda', batch=64); torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
np.random.seed(1); t0 = time.perf_counter()
for _ in range(n): masks.mask_attempt_records(s, [0, 1])
t_rec = time.perf_counter() - t0
np.random.seed(1); t0 = time.perf_counter(); ref = [data.RandomMask(s, [0, 1]) for _ in range(min(n, 64))]; t_host = (time.perf_counter() - t0) / min(n, 64) * n
# rasteriser alone: pre-generated records
np.random.seed(1)
recs, offs, flips = [], [0], []
for _ in range(64):
    r, f0, f1 = masks.mask_attempt_records(s, [0, 1]); recs.append(r); offs.append(offs[-1] + len(r)); flips.append((int(f0), int(f1)))
R = np.concatenate(recs)
masks.rasterize(R, offs, flips, s); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(10): masks.rasterize(R, offs, flips, s)
e1.record(); torch.cuda.synchronize(); t_r = (time.perf_counter() - t0) / 10
print(f's={s}: device path {n / t_dev:8.0f} masks/s (host records alone {n / t_rec:8.0f}/s); host Pillow path {n / t_host:8.0f} masks/s; '
      f'rasterize call (64 masks, H2D + kernel) {t_r * 1e3:.2f} ms = {64 / t_r:.0f} masks/s; equal to host: {bool(np.array_equal(m[:len(ref)].cpu().numpy(), np.stack(ref)))}')
