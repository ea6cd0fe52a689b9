"""One launch each of the round-5 small-layer kernels at training shapes, for a counter pass (tools/gpu_pmc_cmd.sh): thin 1x1 forward and weight
gradient, the Winograd kernels split along the input channels, the stride-2 layer whose split-K plan was fixed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, shgan_amd
from shgan_amd import kernels
d = 'cuda'
for _ in range(3):
    kernels.conv1x1_thin_in(torch.randn(8, 3, 64, 64, device=d), torch.randn(512, 3, device=d), None, act=False)
    kernels.conv1x1_thin_in(torch.randn(16, 4, 512, 512, device=d), torch.randn(64, 4, device=d), None, act=True)
    kernels.conv2d_wgrad(torch.randn(16, 4, 512, 512, device=d), torch.randn(16, 64, 512, 512, device=d), 1, 1, 1, 0)
    kernels.conv2d_wgrad(torch.randn(8, 64, 512, 512, device=d), torch.randn(8, 3, 512, 512, device=d), 1, 1, 1, 0)
    w = torch.randn(512, 512, 3, 3, device=d) / 68
    pw = kernels.conv_weight_prep(w)
    kernels.conv2d(torch.randn(8, 512, 32, 32, device=d), pw, mode=kernels.MODE_SAME, pad=1)
    kernels.conv2d(torch.randn(8, 512, 16, 16, device=d), pw, mode=kernels.MODE_SAME, pad=1)
    kernels.conv2d(torch.randn(8, 512, 65, 65, device=d), pw, mode=kernels.MODE_DOWN2, pad=0)
torch.cuda.synchronize()
