import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import shgan_amd
from shgan_amd import kernels as kk
import torch.nn.functional as F
DEV='cuda:0'
for (n,ci,co,h,w) in [(1,13,70,34,40),(3,24,24,66,36),(2,16,64,32,32)]:
    rs=np.random.RandomState(0)
    x=torch.from_numpy(rs.standard_normal((n,ci,h,w)).astype(np.float32)); wt=torch.from_numpy(rs.standard_normal((co,ci,3,3)).astype(np.float32))
    ref=F.conv_transpose2d(x, wt.transpose(0,1), stride=2)
    pw=kk.conv_weight_prep(wt.to(DEV))
    a=kk.conv2d(x.to(DEV), pw, mode=2, planar=True).cpu()
    for ph in range(4):
        aa,bb=ph>>1,ph&1
        r=ref[:,:,aa::2,bb::2]; g=a[ph][:,:,:h+1-aa,:w+1-bb]
        err=(g-r).abs()
        bad=(err>1e-3*r.abs().max()).nonzero()
        print((n,ci,co,h,w),'phase',ph,'max err',float(err.max()),'nbad',len(bad), 'first', bad[:3].tolist(), 'last', bad[-3:].tolist())
