import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import shgan_amd
from shgan_amd import _lib, kernels as kk
path = '/root/repo/tools/_variants/libshgan_hip_ptt.so'
_lib.use_library(path)
lib = ctypes.CDLL(path)
N, ci, co, h = 16, int(os.environ.get('CI', 64)), int(os.environ.get('CO', 64)), int(os.environ.get('H', 512))
x = torch.randn(N, ci, h, h, device='cuda')
pw = kk.conv_weight_prep(torch.randn(co, ci, 3, 3, device='cuda'))
for _ in range(3):
    y = kk.conv2d(x, pw, mode=0, pad=1, act=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = kk.conv2d(x, pw, mode=0, pad=1, act=True); e1.record(); torch.cuda.synchronize()
buf = np.zeros(1024, dtype=np.int64)
lib.shg_wino4_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
t = buf[512:512 + 60]
t = t[t != 0]
print('kernel %.1f us; tiles recorded %d; total ticks %d -> %.3f ticks/ns' % (e0.elapsed_time(e1) * 1e3, len(t) - 1, t[-1] - t[0], (t[-1] - t[0]) / (e0.elapsed_time(e1) * 1e6)))
print('per-tile ticks:', ' '.join(str(int(v)) for v in np.diff(t)))
