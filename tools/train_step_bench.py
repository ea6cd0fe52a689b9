"""BASELINE config 5 on one GPU: a G + D training step (stylegan_default_loss.py Gmain + Dmain, first order -- no R1 / path-length
regulariser, fp32) of the FFHQ-512 SH-GAN at batch 8, random-init weights, synthetic data.  Times the two phases with HIP events
and prints the per-class kernel times of one step.  usage: python tools/train_step_bench.py [--resolution 512 --batch 8 --steps 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import shgan_amd
from shgan_amd import _lib
if os.environ.get('SHG_VARIANT'):        # A/B: python sh-gan_amd/build.py --variant=<tag> [-D...]
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
from shgan_amd import configs, eval_harness, kernels
from shgan_amd.grad_sync import BucketedAllReduce
from shgan_amd.model_zoo import stylegan

ap = argparse.ArgumentParser()
ap.add_argument('--resolution', type=int, default=512)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--fp16', action='store_true', help='the use_fp16 blocks of the reference: encoder > 64, synthesis > 32, discriminator > 32')
ap.add_argument('--host-enqueue', action='store_true', help='also time how long the host needs to enqueue a G phase (one extra G phase)')
ap.add_argument('--two-pass-critic', action='store_true', help='generated and real batch through the critic separately (A/B)')
ap.add_argument('--direct-convt', action='store_true', help='transposed convolutions on the direct interleaved kernel (A/B)')
a = ap.parse_args()
TWO_PASS_CRITIC = a.two_pass_critic
if a.direct_convt:
    from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix as _gf
    _gf.PLANAR_CONVT = False
dev = 'cuda:0'
G = configs.seeded_init_(configs.build_generator(a.resolution, **(dict(use_fp16_before_res=64, use_fp16_after_res=32) if a.fp16 else {})), seed=0).to(dev).train()
for m in G.modules():                         # dropout of the encoder epilogue stays off: this measures kernels, not RNG
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
D = stylegan.Discriminator(resolution=a.resolution, ic_n=4, ch_base=32768, ch_max=512, use_fp16_before_res=(32 if a.fp16 else None),
                           mbstd_group_size=4, mbstd_c_n=1).to(dev).train()
optG = torch.optim.Adam(G.parameters(), lr=0.002, betas=(0.0, 0.99), eps=1e-8, fused=True)
optD = torch.optim.Adam(D.parameters(), lr=0.002, betas=(0.0, 0.99), eps=1e-8, fused=True)
syncG, syncD = BucketedAllReduce(G.parameters()), BucketedAllReduce(D.parameters())
x, z, _, _ = eval_harness.synthetic_batch(a.batch, a.resolution, 512, seed=1, device=dev, masks='bernoulli')
real = torch.randn(a.batch, 3, a.resolution, a.resolution, device=dev).clamp(-1, 1)
cnd = torch.zeros(a.batch, 0, device=dev)
mask = x[:, 0:1]


def d_in(img):
    return torch.cat([mask, img], dim=1)       # discriminator sees [mask - 0.5, image] (ic_n = 4)


def g_phase():
    G.requires_grad_(True); D.requires_grad_(False)
    syncG.zero_grad()
    with torch.enable_grad():
        img = G(x=x, z=z, c=cnd, noise_mode='random')
        loss = F.softplus(-D(d_in(img), None)).mean()
        loss.backward()
    syncG.finish(); optG.step()
    return float(loss)


def d_phase():
    G.requires_grad_(False); D.requires_grad_(True)
    syncD.zero_grad()
    with torch.no_grad():
        img = G(x=x, z=z, c=cnd, noise_mode='random')
    with torch.enable_grad():
        if TWO_PASS_CRITIC:
            loss = F.softplus(D(d_in(img), None)).mean() + F.softplus(-D(d_in(real), None)).mean()
        else:                               # as losses.StyleGAN2Loss (batch_critic): one pass over the stacked batch, statistic per half
            lg = D(torch.cat([d_in(img), d_in(real)]), None, segments=2)
            loss = F.softplus(lg[:img.shape[0]]).mean() + F.softplus(-lg[img.shape[0]:]).mean()
        loss.backward()
    syncD.finish(); optD.step()
    return float(loss)


hist = []
for _ in range(1):
    hist.append((g_phase(), d_phase()))
torch.cuda.synchronize()
tg = td = 0.0
for _ in range(a.steps):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); lg = g_phase(); e[1].record(); ld = d_phase(); e[2].record(); torch.cuda.synchronize()
    tg += e[0].elapsed_time(e[1]); td += e[1].elapsed_time(e[2]); hist.append((lg, ld))
print(f'G phase {tg / a.steps:8.1f} ms   D phase {td / a.steps:8.1f} ms   step {(tg + td) / a.steps:8.1f} ms   '
      f'({a.batch / ((tg + td) / a.steps) * 1e3:.1f} images/s, losses {lg:.4f} {ld:.4f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)')
if a.host_enqueue:
    # host time to ENQUEUE a step (the losses are read with float(): one sync per phase, so this is measured on the autograd part alone)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    G.requires_grad_(True); D.requires_grad_(False); syncG.zero_grad()
    with torch.enable_grad():
        _img = G(x=x, z=z, c=cnd, noise_mode='random')
        _loss = F.softplus(-D(d_in(_img), None)).mean()
        _loss.backward()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host enqueue of the G phase (forward + backward, no sync): {(t1 - t0) * 1e3:.1f} ms; GPU drained {(t2 - t1) * 1e3:.1f} ms later')
print('loss history (G, D) per step:', ' '.join(f'({g:.4f} {d:.4f})' for g, d in hist))
kt = kernels.KernelTimer()
kernels.set_timer(kt)
g_phase(); d_phase()
kernels.set_timer(None)
torch.cuda.synchronize()
tot = kt.summary()
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]['ms'])[:16]:
    print(f'   {k:16s} {v["ms"]:8.2f} ms  {v["calls"]:4d} launches' + (f'  {v["work"] / v["ms"] / 1e9:7.1f} TFLOP/s' if k.startswith('conv') else ''))
