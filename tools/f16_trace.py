"""Workgroup timelines of conv_f16_kernel (variant built with `python sh-gan_amd/build.py --variant=f16trace -DSHG_F16_TRACE=1`): every 61st
workgroup records clock64() at entry | first chunk in LDS | first chunk multiplied | all chunks multiplied | stores issued | stores drained.
usage: CI=64 CO=64 H=512 python tools/f16_trace.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import shgan_amd
from shgan_amd import _lib
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ.get('SHG_VARIANT', 'f16trace'))
_lib.use_library(path)
from shgan_amd import kernels_f16 as kf
lib = ctypes.CDLL(path)
N, ci, co, h = int(os.environ.get('NB', 8)), int(os.environ.get('CI', 64)), int(os.environ.get('CO', 64)), int(os.environ.get('H', 512))
x = torch.randn(N, ci, h, h, device='cuda').half().to(memory_format=torch.channels_last)
w = (torch.randn(co, ci, 3, 3, device='cuda') / (ci * 9) ** 0.5).half()
for _ in range(3):
    y = kf.conv2d(x, w, None, 1, 1)
torch.cuda.synchronize()
buf = np.zeros(256 * 8, dtype=np.int64)
assert lib.shg_f16_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
R = buf.reshape(256, 8)
R = R[R[:, 0] > 0]
T = buf.reshape(256, 8)[:, :6]
T = T[T[:, 0] > 0]
t0 = T[:, 0].min()
print(f'{len(T)} traced workgroups; clock64 ticks (100 MHz: 1 tick = 10 ns) relative to the earliest entry')
print('   wg   entry | +LDS0 | +mul0 | +mul all | +stores issued | +drained | lifetime')
for k, r in enumerate(T[:: max(1, len(T) // 40)]):
    d = np.diff(r)
    print(f'{k:5d} {int(r[0] - t0):7d} | ' + ' | '.join(f'{int(v):6d}' for v in d) + f' | {int(r[5] - r[0]):6d}')
d = np.diff(T, axis=1)
print('median phase ticks:', ' '.join(f'{int(np.median(d[:, k])):6d}' for k in range(5)), ' lifetime', int(np.median(T[:, 5] - T[:, 0])), ' span', int(T[:, 5].max() - t0))
print('epilogue split, median ticks: all waves done multiplying (barrier)', int(np.median(R[:, 6] - R[:, 3])), '| tile written to LDS + barrier', int(np.median(R[:, 7] - R[:, 6])), '| read back + stores issued', int(np.median(R[:, 4] - R[:, 7])))
