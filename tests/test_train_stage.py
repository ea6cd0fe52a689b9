"""Training stage (shgan_amd.train_stage = lib/experiments/stylegan_default.py:108-167, :304-321, :383-390): host logic on CPU with a
stand-in loss, two real iterations on the GPU."""
import copy

import numpy as np
import pytest
import torch

import shgan_amd  # noqa: F401
from shgan_amd import train_stage as ts


class _StubLoss:
    """Records the calls; gives every parameter of the module being trained the gradient `gain` (+ one NaN to be sanitised)."""

    def __init__(self, G, D):
        self.G, self.D, self.calls = G, D, []

    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync, gain):
        self.calls.append((phase, tuple(real_img.shape), tuple(gen_z.shape), bool(sync), gain))
        mod = self.G if phase.startswith('G') else self.D
        assert all(p.requires_grad for p in mod.parameters())
        other = self.D if phase.startswith('G') else self.G
        assert not any(p.requires_grad for p in other.parameters())
        with torch.enable_grad():       # (conftest runs the tests under no_grad: inference is the default path)
            out = sum((p * float(gain)).sum() for p in mod.parameters())
            out.backward()
        if phase == 'Dmain':
            next(iter(mod.parameters())).grad.view(-1)[0] = float('nan')


def test_phases_lazy_regularisation_and_schedule():
    G, D = torch.nn.Linear(3, 2).requires_grad_(False), torch.nn.Linear(2, 1).requires_grad_(False)     # (stylegan_default.py:229-230)
    kw = dict(lr=0.002, betas=(0.0, 0.99), eps=1e-8)
    phases = ts.make_phases(G, D, kw, kw, g_reg_interval=4, d_reg_interval=16)
    assert [(p.name, p.interval) for p in phases] == [('Gmain', 1), ('Greg', 4), ('Dmain', 1), ('Dreg', 16)]
    assert phases[0].opt is phases[1].opt and phases[2].opt is phases[3].opt
    g = phases[0].opt.param_groups[0]
    assert abs(g['lr'] - 0.002 * 4 / 5) < 1e-12 and abs(g['betas'][1] - 0.99 ** (4 / 5)) < 1e-12 and g['betas'][0] == 0.0
    d = phases[2].opt.param_groups[0]
    assert abs(d['lr'] - 0.002 * 16 / 17) < 1e-12 and abs(d['betas'][1] - 0.99 ** (16 / 17)) < 1e-12
    both = ts.make_phases(G, D, kw, kw, g_reg_interval=None, d_reg_interval=None)
    assert [(p.name, p.interval) for p in both] == [('Gboth', 1), ('Dboth', 1)] and both[0].opt.param_groups[0]['lr'] == 0.002
    loss = _StubLoss(G, D)
    real = torch.randn(4, 3, 8, 8)
    ran = [ts.run_phases(real, 5, phases, i, loss, batch_gpu=4, effective_batch_gpu=2) for i in range(17)]
    assert ran[0] == ['Gmain', 'Greg', 'Dmain', 'Dreg'] and ran[1] == ['Gmain', 'Dmain'] and ran[4] == ['Gmain', 'Greg', 'Dmain']
    assert ran[16] == ['Gmain', 'Greg', 'Dmain', 'Dreg']
    # two rounds of effective_batch_gpu samples per phase, sync on the last, gain = interval
    first = loss.calls[:8]
    assert first[0] == ('Gmain', (2, 3, 8, 8), (2, 5), False, 1) and first[1] == ('Gmain', (2, 3, 8, 8), (2, 5), True, 1)
    assert first[2][0] == 'Greg' and first[2][4] == 4 and first[6][0] == 'Dreg' and first[6][4] == 16
    assert all(torch.isfinite(p).all() for p in D.parameters())          # the NaN gradient was sanitised before the step
    assert not any(p.requires_grad for p in list(G.parameters()) + list(D.parameters()))


def test_ema_beta_and_update():
    assert abs(ts.ema_beta(32, 10 ** 9, ema_kimg=10) - 0.5 ** (32 / 10000)) < 1e-15
    assert abs(ts.ema_beta(32, 2000, ema_kimg=10, ema_rampup=0.05) - 0.5 ** (32 / 100)) < 1e-15
    assert ts.ema_beta(32, 0, ema_kimg=10, ema_rampup=0.05) == 0.0          # first iteration: G_ema = G
    G = torch.nn.BatchNorm1d(3)
    G_ema = copy.deepcopy(G)
    with torch.no_grad():
        G.weight.fill_(3.0); G.running_mean.fill_(7.0)
    beta = ts.update_ema(G_ema, G, batch_size=8, cur_nimg=10 ** 9, ema_kimg=0.016)
    assert abs(beta - 0.5 ** 0.5) < 1e-12
    assert torch.allclose(G_ema.weight, torch.full((3,), 3.0 + (1.0 - 3.0) * beta)) and torch.equal(G_ema.running_mean, G.running_mean)


def test_train_loop_counts():
    G, D = torch.nn.Linear(3, 2).requires_grad_(False), torch.nn.Linear(2, 1).requires_grad_(False)
    kw = dict(lr=0.002, betas=(0.0, 0.99), eps=1e-8)
    phases = ts.make_phases(G, D, kw, kw)
    ticks = []
    n, idx = ts.train(G, D, copy.deepcopy(G), _StubLoss(G, D), (torch.randn(2, 3, 4, 4) for _ in range(100)), phases, z_dim=5,
                      batch_size=4, batch_gpu=2, total_kimg=0.02, kimg_per_tick=0.008, on_tick=lambda t, nimg, i: ticks.append((t, nimg)))
    assert (n, idx) == (20, 5) and ticks == [(0, 4), (1, 12), (2, 20)]


@pytest.mark.gpu
def test_two_training_iterations_on_the_gpu():
    """Plain StyleGAN2 G / D at reduced width, StyleGAN2Loss, all four phases twice: parameters move, stay finite, only the phase's
    network is touched, G_ema follows."""
    from shgan_amd import losses
    from shgan_amd.model_zoo import stylegan as sg
    dev = 'cuda:0'
    torch.manual_seed(5)
    mp = sg.Mapping(z_dim=32, c_dim=0, w_dim=32, num_ws=8, num_layers=2, lr_multiplier=0.01, w_avg_beta=0.995)
    syn = sg.Synthesis(w_dim=32, resolution=32, rgb_n=3, ch_base=256, ch_max=16, use_fp16_after_res=32)
    G = sg.Generator(mp, syn).to(dev).train().requires_grad_(False)
    D = sg.Discriminator(resolution=32, ic_n=3, ch_base=256, ch_max=16, use_fp16_before_res=None, mbstd_group_size=4,
                         mbstd_c_n=1).to(dev).train().requires_grad_(False)
    G_ema = copy.deepcopy(G).eval()
    kw = dict(lr=0.002, betas=(0.0, 0.99), eps=1e-8)
    phases = ts.make_phases(G, D, kw, kw, g_reg_interval=4, d_reg_interval=16)
    assert len(phases[0].sync.params) == len(list(G.parameters())) and len(phases[2].sync.params) == len(list(D.parameters()))
    loss = losses.StyleGAN2Loss(dev, G.mapping, G.synthesis, D, style_mixing_prob=0.9, r1_gamma=10, pl_batch_shrink=2)
    g0 = [p.detach().clone() for p in G.parameters()]
    d0 = [p.detach().clone() for p in D.parameters()]
    batches = (torch.randn(4, 3, 32, 32, device=dev).clamp(-1, 1) for _ in range(2))
    n, idx = ts.train(G, D, G_ema, loss, batches, phases, z_dim=32, batch_size=4, batch_gpu=4, total_kimg=1, ema_kimg=0.004)
    assert (n, idx) == (8, 2)
    assert all(torch.isfinite(p).all() for p in list(G.parameters()) + list(D.parameters()))
    moved_g = sum(int((p - q).abs().max() > 0) for p, q in zip(G.parameters(), g0))
    moved_d = sum(int((p - q).abs().max() > 0) for p, q in zip(D.parameters(), d0))
    assert moved_g >= len(g0) - 2 and moved_d >= len(d0) - 2
    beta = ts.ema_beta(4, 4, ema_kimg=0.004)
    last = list(G.parameters())[0]
    assert torch.isfinite(list(G_ema.parameters())[0]).all() and 0 < beta < 1 and last.shape == list(G_ema.parameters())[0].shape


def test_gloo_world2_training_stage_keeps_ranks_identical():
    """Two gloo ranks, different data per rank, a hand-written least-squares 'GAN': after ``train`` both ranks hold bit-identical
    parameters, equal to a one-process run on the mean gradient; a parameter the reg phase never touches is not stepped by it."""
    import os, subprocess, sys
    from conftest import ROOT
    script = r'''
import copy, os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd import train_stage as ts
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"], rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
def nets():
    torch.manual_seed(0)
    G = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 12)).requires_grad_(False)
    D = torch.nn.Sequential(torch.nn.Linear(12, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1)).requires_grad_(False)
    return G, D
class Loss:
    def __init__(self, G, D, world_mean):
        self.G, self.D, self.wm, self.grad_sync = G, D, world_mean, None
    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync, gain):
        with torch.enable_grad():
            if phase == "Gmain":   l = torch.nn.functional.softplus(-self.D(self.G(gen_z))).mean()
            elif phase == "Greg":  l = self.G[0](gen_z).square().mean()                      # touches only the first layer
            elif phase == "Dmain":                                                            # TWO backward passes, like Dgen + Dreal
                (torch.nn.functional.softplus(self.D(self.G(gen_z).detach())).mean() * gain / self.wm).backward()
                l = torch.nn.functional.softplus(-self.D(real_img.flatten(1))).mean()
            else:                  l = self.D(real_img.flatten(1)).square().mean()
            if sync and self.grad_sync is not None: self.grad_sync.arm()
            (l * gain / self.wm).backward()
kw = dict(lr=0.01, betas=(0.0, 0.99), eps=1e-8)
def run(rank_data, world_mean, G, D):
    torch.manual_seed(123)                       # same latents on every rank / in the reference run
    phases = ts.make_phases(G, D, kw, kw, g_reg_interval=2, d_reg_interval=3)
    return ts.train(G, D, copy.deepcopy(G), Loss(G, D, world_mean), iter(rank_data), phases, z_dim=6, batch_size=8, batch_gpu=4, total_kimg=1, effective_batch_gpu=2)
data = [[torch.randn(4, 3, 2, 2, generator=torch.Generator().manual_seed(100 * k + it)) for it in range(4)] for k in range(2)]
G, D = nets()
run(data[r], 1.0, G, D)
flat = torch.cat([p.reshape(-1) for p in list(G.parameters()) + list(D.parameters())])
both = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1])
dist.destroy_process_group()
print("rank", r, "ok", float(flat.abs().sum()))
'''
    port = str(35500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), SHG_ROOT=ROOT, SHG_PORT=port)
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f'rank {rank} ok' in o, o
    assert outs[0].split()[-1] == outs[1].split()[-1]
