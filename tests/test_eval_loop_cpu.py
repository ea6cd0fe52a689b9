"""CPU (gloo, world size 2): the streamed evaluation loop ``eval_harness.EvalLoop`` -- lib/experiments/shgan_default.py:264-300 with
the collectives of SURVEY 8(e): per-batch nothing, at the end one all-gather of the uint8 results + zipzap and one all-reduce of the
FID moments -- and ``broadcast_state``, the rank-0 checkpoint load + weight broadcast of shgan_default.py:138-154,223-231.  The
generator step and the moment kernel need a GPU; stand-ins are injected for both (the product forms run in tests/test_gpu_eval_loop.py)."""
import os
import subprocess
import sys

from conftest import ROOT


def _run_two(script, port_base):
    port = str(port_base + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), SHG_ROOT=ROOT, SHG_PORT=port)
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()


def test_gloo_world2_eval_loop_streams_gathers_and_reduces_moments():
    script = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd import eval_harness as hz, data
from shgan_amd.fid_stats import FidStats
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"], rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
R, N, B, D = 32, 11, 4, 16
def step(x, z, out):                       # stand-in generator step writing INTO the loop's result buffer
    img = torch.tanh(x[:, 1:4] * 0.5 + z[:, :3, None, None] * 0.1)
    m = x[:, 0:1] + 0.5
    out.copy_(((x[:, 1:4] * m + img * (1 - m)) * 127.5 + 127.5).clamp(0, 255).to(torch.uint8))
    return out
def acc(S, feats, w):                      # stand-in for the fp64-MFMA moment kernel: the same augmented second moments
    f = torch.cat([feats.double(), torch.ones(feats.shape[0], 1, dtype=torch.float64)], 1)
    if w is not None:
        S[:D + 1, :D + 1] += (f * w.double()[:, None]).t() @ f
    else:
        S[:D + 1, :D + 1] += f.t() @ f
def latents(ids, b):
    g = torch.Generator()
    out = torch.empty(b, 8)
    for k, i in enumerate(ids):
        g.manual_seed(100 + int(i)); out[k].normal_(generator=g)
    return out
def masks_for(ids):
    return torch.stack([((torch.arange(R * R).reshape(R, R) * (int(i) + 3)) % 7 > 2).float() for i in ids])
class Loader:
    def __init__(self, ids): self.ids = ids
    def __iter__(self):
        inner = hz.PinnedU8Loader(self.ids, B, R, seed=5)
        for img, ids in inner:
            yield img, masks_for(ids), ids
feat = lambda u8: hz.standin_features(u8, D)
def run(rank, world):
    loop = hz.EvalLoop(None, "cpu", R, N, rank=rank, world=world, noise_mode="const", feature_fn=feat, fid_dim=D, latent_fn=latents,
                       device_masks=False, step_fn=step, fid_accumulate_fn=acc)
    seen = []
    loop.on_batch = lambda ids, out, ev: seen.append(list(ids))
    loop.run(Loader(loop.ids))
    assert sum(len(s) for s in seen) == len(loop.ids) and [i for s in seen for i in s] == loop.ids
    return loop
loop = run(r, 2)
assert loop.ids == ([0, 2, 4, 6, 8, 10] if r == 0 else [1, 3, 5, 7, 9, 0])
images, fid = loop.gather()
assert images.shape == (N, 3, R, R) and images.dtype == torch.uint8
# the same evaluation on one rank without a process group's help: built by hand from the pieces
one = hz.EvalLoop(None, "cpu", R, N, rank=0, world=1, noise_mode="const", feature_fn=feat, fid_dim=D, latent_fn=latents, device_masks=False,
                  step_fn=step, fid_accumulate_fn=acc)
one.run(Loader(one.ids))
assert torch.equal(images, one.images), "gathered + zipzapped result != the 1-rank run"
# zipzap on the device == the reference's host re-interleave
full = torch.stack([run(q, 2).images for q in range(2)])
assert np.array_equal(hz.zipzap_device(full, N).numpy(), data.zipzap_arrange([full[0].numpy(), full[1].numpy()])[:N])
# moments: the padded duplicate (item 0 on rank 1) has weight 0, so the all-reduced sum equals the 1-rank sum and counts N samples
n2, mu2, sg2 = fid.mean_cov()
n1, mu1, sg1 = one.local_fid().mean_cov()
assert n2 == n1 == N, (n2, n1)
assert np.allclose(mu2, mu1, rtol=0, atol=1e-12) and np.allclose(sg2, sg1, rtol=0, atol=1e-9)
dist.destroy_process_group()
print("rank", r, "ok")
'''
    _run_two(script, 33500)


def test_gloo_world2_broadcast_state_from_rank0():
    script = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd import eval_harness as hz
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"], rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(7, 5)
        self.b = torch.nn.Conv2d(3, 4, 3)
        self.register_buffer("avg", torch.zeros(5))
        self.register_buffer("steps", torch.zeros((), dtype=torch.int64))
torch.manual_seed(10 + r)                  # the ranks start DIFFERENT: only rank 0 "read the checkpoint"
net = Net()
if r == 0:
    net.avg.fill_(0.25); net.steps.fill_(12345)
ref = Net(); torch.manual_seed(10); ref2 = Net(); ref2.avg.fill_(0.25); ref2.steps.fill_(12345)
v0 = {k: p._version for k, p in net.named_parameters()}
nbytes = hz.broadcast_state(net, src=0)
want = sum(t.numel() * 4 for t in list(net.parameters()) + [net.avg]) + 8
assert nbytes == want, (nbytes, want)
for (k, t), (_, u) in zip(sorted(net.state_dict().items()), sorted(ref2.state_dict().items())):
    assert t.dtype == u.dtype and torch.equal(t, u), k
assert all(p._version > v0[k] for k, p in net.named_parameters()), "parameters must be written through copy_ (version counters move)"
dist.destroy_process_group()
print("rank", r, "ok")
'''
    _run_two(script, 35500)


def test_broadcast_state_without_a_process_group_is_a_no_op():
    import torch
    import shgan_amd  # noqa: F401
    from shgan_amd import eval_harness as hz
    net = torch.nn.Linear(3, 2)
    w = net.weight.detach().clone()
    assert hz.broadcast_state(net) == 0 and torch.equal(net.weight, w)
