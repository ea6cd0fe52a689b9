import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.nn.functional as F
import shgan_amd
from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix as gf
from shgan_amd import kernels
DEV='cuda:0'
def rel(a,b): return float(np.abs(a-b).max()/np.abs(b).max())
rs=np.random.RandomState(0)
for (n,ci,co,h,w,pad) in [(2,256,128,64,64,0),(2,128,64,32,32,1),(1,64,64,16,16,0),(2,512,512,32,32,0),(2,64,3,64,64,0)]:
    x=torch.from_numpy(rs.standard_normal((n,ci,h,w)).astype(np.float32)); wt=torch.from_numpy((rs.standard_normal((ci,co,3,3))/np.sqrt(ci*9)).astype(np.float32))
    b=torch.from_numpy(rs.standard_normal(co).astype(np.float32))
    ref=F.conv_transpose2d(x.double(),wt.double(),b.double(),stride=2,padding=pad).numpy()
    for planar in (True, False):
        gf.PLANAR_CONVT=planar
        y=gf.conv_transpose2d(x.to(DEV),wt.to(DEV),b.to(DEV),stride=2,padding=pad).cpu().numpy()
        print('convT',(n,ci,co,h,w,pad),'planar',planar,rel(y,ref), kernels._lib.get_lib().shg_conv2d_up_poly_supported(n,ci,co,h,w))
# input gradient of stride-2 conv
for (n,ci,co,h,w,pad) in [(2,128,256,129,129,0),(2,64,128,257,257,0),(2,512,512,65,65,0),(2,32,64,64,64,1)]:
    x=torch.from_numpy(rs.standard_normal((n,ci,h,w)).astype(np.float32)); wt=torch.from_numpy((rs.standard_normal((co,ci,3,3))/np.sqrt(ci*9)).astype(np.float32))
    xr=x.double().requires_grad_(); yr=F.conv2d(xr,wt.double(),stride=2,padding=pad); gy=torch.from_numpy(rs.standard_normal(tuple(yr.shape)).astype(np.float32)); yr.backward(gy.double())
    for planar in (True, False):
        gf.PLANAR_CONVT=planar
        xd=x.to(DEV).requires_grad_(); wd=wt.to(DEV).requires_grad_()
        with torch.enable_grad():
            y=gf.conv2d(xd,wd,None,stride=2,padding=pad); y.backward(gy.to(DEV))
        print('dgrad',(n,ci,co,h,w,pad),'planar',planar,rel(xd.grad.cpu().numpy(),xr.grad.numpy()))
