"""Is the run-to-run round-off difference of the four-phase training iteration (second-order path-length / R1 passes) a property of
torch's autograd ENGINE rather than of a kernel?  Nodes created during a create_graph backward get their sequence numbers from the
device worker thread's counter, forward nodes from the main thread's; the two counters advance by different amounts per iteration, so
the ready-queue order of the double-backward graph -- and with it the order in which multi-consumer gradients (x_global, the skip
features) are accumulated -- drifts between executions.  With torch.autograd.set_multithreading_enabled(False) (train_stage.SINGLE_THREADED_BACKWARD, the
product's setting) every node is numbered by one thread.  usage: python tools/probes/autograd_thread_order.py [--fp16]"""
import os, sys, copy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import shgan_amd
from shgan_amd import losses, train_stage as ts
from test_gpu_config5 import build_networks
DEV = torch.device('cuda:0')
fp16 = '--fp16' in sys.argv
G, D = build_networks(512, 61, 62, fp16=fp16)
G.requires_grad_(False); D.requires_grad_(False)
g0, d0 = copy.deepcopy(G.state_dict()), copy.deepcopy(D.state_dict())
rs = np.random.RandomState(63)
real = torch.from_numpy(rs.uniform(-1, 1, size=(8, 3, 512, 512)).astype(np.float32))
mask = torch.from_numpy((rs.uniform(size=(8, 1, 512, 512)) < 0.7).astype(np.float32))
real4 = torch.cat([mask - 0.5, real], dim=1).to(DEV)
kw = dict(lr=0.002, betas=(0.0, 0.99), eps=1e-8)


def iteration():
    G.load_state_dict(g0); D.load_state_dict(d0)
    torch.manual_seed(7)
    L = losses.InpaintingLoss(DEV, G, D, noise_mode='const', style_mixing_prob=0.9)
    phases = ts.make_phases(G, D, kw, kw, g_reg_interval=4, d_reg_interval=16)
    ts.run_phases(real4, 512, phases, batch_idx=0, loss=L, batch_gpu=8, device=DEV)
    out = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())]).clone()
    for ph in phases:
        if ph.sync is not None:
            ph.sync.remove()
    return out


small = '--small' in sys.argv
if small:
    from test_gpu_train_graph import small_networks, real_batch
    G, D = small_networks(5)
    g0, d0 = copy.deepcopy(G.state_dict()), copy.deepcopy(D.state_dict())
    real4 = real_batch(4, 6)

    def iteration():
        G.load_state_dict(g0); D.load_state_dict(d0)
        torch.manual_seed(11)
        L = losses.InpaintingLoss(DEV, G, D, noise_mode='const', style_mixing_prob=0)
        phases = ts.make_phases(G, D, kw, kw, g_reg_interval=4, d_reg_interval=16)
        ts.run_phases(real4, 64, phases, batch_idx=0, loss=L, batch_gpu=4, device=DEV)
        out = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())]).clone()
        for ph in phases:
            if ph.sync is not None:
                ph.sync.remove()
        return out

# one mode per process (the first executions of a process are the interesting ones): SHG_ENGINE_THREADS=1 selects torch's default
runs = [iteration() for _ in range(5)]
print(f'single-threaded backward = {ts.SINGLE_THREADED_BACKWARD}, {"reduced-width 256" if small else "FFHQ-512"}: differing elements of run k vs run 0:',
      [int((runs[0] != r).sum()) for r in runs[1:]], ' run k vs run 1:', [int((runs[1] != r).sum()) for r in runs[2:]], flush=True)
