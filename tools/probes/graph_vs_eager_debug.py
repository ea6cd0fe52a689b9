import os, sys, copy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from test_gpu_train_graph import small_networks, real_batch, DEV
from shgan_amd import losses, train_stage as ts
G, D = small_networks(5)
g0, d0 = copy.deepcopy(G.state_dict()), copy.deepcopy(D.state_dict())
real4 = real_batch(4, 6)
kw = dict(lr=0.002, betas=(0.0, 0.99), eps=1e-8, capturable=True)
for order in ([1], [1, 1], [1, 1, 1], [1, 1, 1, 1], [0], [0, 4], [0, 4, 8], [0, 4, 8, 12]):
    out = []
    for graphed in (False, True):
        G.load_state_dict(g0); D.load_state_dict(d0)
        torch.manual_seed(11)
        L = losses.InpaintingLoss(DEV, G, D, noise_mode='const', style_mixing_prob=0)
        phases = ts.make_phases(G, D, kw, kw, g_reg_interval=4, d_reg_interval=16)
        pg = ts.PhaseGraphs(phases, L, 4, 64, tuple(real4.shape), DEV) if graphed else None
        for idx in order:
            pg.run(real4, idx) if graphed else ts.run_phases(real4, 64, phases, batch_idx=idx, loss=L, batch_gpu=4, device=DEV)
        torch.cuda.synchronize()
        out.append({n: p.detach().clone() for n, p in list(G.named_parameters()) + [('D.' + n, p) for n, p in D.named_parameters()]})
        for ph in phases:
            if ph.sync is not None:
                ph.sync.remove()
    bad = sorted(((float((out[0][n] - out[1][n]).abs().max()), n) for n in out[0]), reverse=True)
    print(order, 'graphs' , sorted(pg.graphs), 'differing tensors', sum(1 for b in bad if b[0] > 0), bad[:4], flush=True)
