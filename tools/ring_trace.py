"""Step timeline of the fp16 ring convolution (csrc/conv_f16_ring.hip, variant built with -DSHG_RING_TRACE=1): cycles per step of workgroup 0, waves 0 and 7:
wait (vmcnt(0) + barrier) | issue (next step's DMA requests, deferred stores) | multiply (72 MFMAs per wave, + pack on a tile's last chunk).
usage: python sh-gan_amd/build.py --variant=ringtrace -DSHG_RING_TRACE=1 && python tools/ring_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import shgan_amd
from shgan_amd import _lib
_lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_ringtrace.so'))
from shgan_amd import kernels_f16 as kf
dev = 'cuda:0'
for (i, o, r) in [(64, 64, 512), (128, 128, 256), (512, 512, 64)]:
    x = torch.randn(8, i, r, r, device=dev).half().to(memory_format=torch.channels_last)
    w = (torch.randn(o, i, 3, 3, device=dev) / (i * 9) ** 0.5).half()
    for _ in range(3):
        kf.conv2d(x, w, None, 1, 1)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (2 * 64 * 4))()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    assert lib.shg_ring_trace_read(buf) == 0
    t = np.array(buf, dtype=np.int64).reshape(2, 64, 4)
    n = 32
    print(f'--- {i}->{o} {r}^2 batch 8: cycles (s_memtime, 100 MHz-class counter scaled? raw ticks) per step, first {n} steps')
    for wv, name in ((0, 'wave 0'), (1, 'wave 7')):
        tt = t[wv, :n]
        wait, issue, mul = tt[:, 1] - tt[:, 0], tt[:, 2] - tt[:, 1], tt[:, 3] - tt[:, 2]
        period = np.diff(tt[:, 0])
        print(f'{name}: wait    ', ' '.join(f'{v:5d}' for v in wait))
        print(f'{name}: issue   ', ' '.join(f'{v:5d}' for v in issue))
        print(f'{name}: multiply', ' '.join(f'{v:5d}' for v in mul))
        print(f'{name}: period  ', ' '.join(f'{v:5d}' for v in period), '| median', int(np.median(period)))
