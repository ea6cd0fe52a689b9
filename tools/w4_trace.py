"""Per-wave timeline of the F(4x4) kernel's main loop (variant built with -DSHG_W4_TRACE=1): workgroup 0, chunks 8..15."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import shgan_amd
from shgan_amd import _lib, kernels as kk
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ.get('SHG_VARIANT', 'trace'))
_lib.use_library(path)
lib = ctypes.CDLL(path)
N, ci, co, h = int(os.environ.get("NB", 16)), int(os.environ.get('CI', 512)), int(os.environ.get('CO', 512)), int(os.environ.get('H', 64))
x = torch.randn(N, ci, h, h, device='cuda')
pw = kk.conv_weight_prep(torch.randn(co, ci, 3, 3, device='cuda'))
for _ in range(3):
    y = kk.conv2d(x, pw, mode=0, pad=1, act=True)
torch.cuda.synchronize()
buf = np.zeros(8 * 8 * 8, dtype=np.int64)
assert lib.shg_wino4_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
T = buf.reshape(8, 8, 8)[:, 7, :]
t = buf.reshape(8, 8, 8)[:, :7, :int(os.environ.get('NSLOT', 7))]
t0 = t[:, 0, 0].min()
print('clock64 ticks relative to the first traced barrier release; columns: start | after carried group | ks0 | ks1 | ks2 done | before barrier | after barrier')
for c in range(7):
    for w in range(8):
        print(f'chunk {c} wave {w} ({"X" if w < 4 else "L"}): ' + ' '.join(f'{int(v - t0):7d}' for v in t[w, c]))
    print()
d = t[:, 1:, 0] - t[:, :-1, 0]
print('tile phases (ticks from entry): prologue done | main loop done | pass 0..3 done')
for w in range(8):
    print(f'wave {w}: ' + ' '.join(f'{int(v - T[w, 0]):7d}' for v in T[w, 1:7]))
print('chunk period (ticks):', d.mean(), ' -> multiply by core/clock64 ratio')

if os.environ.get('EPI'):
    E = buf.reshape(8, 64)
    print('epilogue passes (ticks from pass 0 start): start | LDS written + barrier passed | A^T.A rows done | tail math done | next loads issued | stores issued')
    for w in (0, 4):
        for ps in range(4):
            print(f'wave {w} pass {ps}: ' + ' '.join(f'{int(E[w, ps * 8 + k] - E[w, 0]):7d}' for k in (0, 1, 2, 4, 5, 3)))
