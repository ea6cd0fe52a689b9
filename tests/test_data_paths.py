"""CPU: the product's integer paths (masks, sampler, re-interleave) are bit-exact with the reference's
golden outputs; world_size-2 gloo run of the sharded index logic."""
import hashlib
import os
import subprocess
import sys

import numpy as np

import shgan_amd  # noqa: F401
from conftest import ROOT, load_golden
from shgan_amd import data


def test_random_mask_bit_exact():
    g = load_golden('integer_paths')
    for s in (64, 256, 512):
        np.random.seed(0)
        for i in range(4):
            m = data.RandomMask(s, [0, 1])
            assert m.shape == (1, s, s) and m.dtype == np.float32 and set(np.unique(m)) <= {0.0, 1.0}
            bits = np.packbits(m.astype(np.uint8))
            assert np.array_equal(bits, g[f'mask{s}_bits'][i]), (s, i)
            assert hashlib.sha256(bits.tobytes()).hexdigest()[:16] == str(g[f'mask{s}_sha'][i])


def test_sampler_and_zipzap_bit_exact():
    g = load_golden('integer_paths')
    for row in g['sampler_rows']:
        n_items, world, rank = [int(v) for v in row[:3]]
        smp = data.DistributedSampler(list(range(n_items)), num_replicas=world, rank=rank, shuffle=False, extend=True)
        assert list(iter(smp)) == [int(v) for v in row[3:]]    # incl. the short-pad quirk when n_items < world/2
    assert data.zipzap_arrange([[0, 2, 4, 6], [1, 3, 5, 7]]) == list(g['zipzap_out'])
    assert data.zipzap_arrange([[0, 3, 6], [1, 4, 7], [2, 5]]) == list(g['zipzap_out_ragged'])
    a = data.zipzap_arrange([np.arange(0, 8, 2).reshape(4, 1), np.arange(1, 6, 2).reshape(3, 1)])
    assert a[:, 0].tolist() == [0, 1, 2, 3, 4, 5, 6]
    # empty / ragged edge cases
    assert list(iter(data.DistributedSampler([], num_replicas=2, rank=1, shuffle=False, extend=True))) == []
    # reference quirk kept on purpose: the pad re-uses at most len(dataset) leading ids, so with 1 item and
    # 4 ranks only ranks 0 and 1 receive a sample (ds_sampler.py:60-62)
    got = [list(iter(data.DistributedSampler([0], num_replicas=4, rank=r, shuffle=False, extend=True))) for r in range(4)]
    assert got == [[0], [0], [], []]


def test_gloo_world2_sharding_covers_dataset_once():
    """N > 1 path on CPU: two gloo ranks shard 11 items, all-gather their ids, re-interleave."""
    script = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd import data
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
ids = list(iter(data.DistributedSampler(list(range(11)), shuffle=False, extend=True)))
t = torch.tensor(ids)
out = [torch.zeros_like(t) for _ in range(2)]
dist.all_gather(out, t)
order = data.zipzap_arrange([o.tolist() for o in out])
assert order[:11] == list(range(11)), order
smp = data.DistributedSampler(list(range(11)), shuffle=True, extend=True)
a = torch.tensor(list(iter(smp)))
g = [torch.zeros_like(a) for _ in range(2)]
dist.all_gather(g, a)
assert sorted(data.zipzap_arrange([x.tolist() for x in g])[:11]) == list(range(11))
dist.destroy_process_group()
print("rank", r, "ok")
'''
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), SHG_ROOT=ROOT, SHG_PORT=port)
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_gloo_world2_sharded_eval_gather_and_zipzap():
    """BASELINE config 4's loop on CPU: two gloo ranks run ``eval_harness.sharded_eval`` end to end -- rank-strided ids,
    batched per-item input synthesis, the all_gather_into_tensor + zipzap re-interleave branch -- with a stand-in for the
    generator step (the product step needs a GPU); the gathered result must equal the 1-rank run item for item, and the
    padded tail (11 items over 2 ranks -> 12 slots) must be dropped (ds_sampler.py:60-62, eva_base.py:196-230)."""
    script = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd import eval_harness as hz
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
def step(x, z):      # stand-in generator: per-item function of (x, z) -> uint8 [B,3,R,R]
    img = torch.tanh(x[:, 1:4] * 0.5 + z[:, :3, None, None] * 0.1)
    m = x[:, 0:1] + 0.5
    return ((x[:, 1:4] * m + img * (1 - m)) * 127.5 + 127.5).clamp(0, 255).to(torch.uint8)
kw = dict(n_items=11, batch_size=4, resolution=32, seed=7, device="cpu", step_fn=step, z_dim=8)
order, merged = hz.sharded_eval(None, rank=r, world=2, gather=True, **kw)
ids1, out1 = hz.sharded_eval(None, rank=0, world=1, gather=False, **kw)
assert order == ids1 == list(range(11)), order
assert merged.shape == (11, 3, 32, 32) and merged.dtype == np.uint8
assert np.array_equal(merged, out1.numpy())
ids_r, local = hz.sharded_eval(None, rank=r, world=2, gather=False, **kw)
assert ids_r == ([0, 2, 4, 6, 8, 10] if r == 0 else [1, 3, 5, 7, 9, 0]), ids_r      # extend-pad re-uses item 0
assert np.array_equal(local.numpy()[:5], merged[r:10:2])
dist.destroy_process_group()
print("rank", r, "ok")
'''
    port = str(31500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), SHG_ROOT=ROOT, SHG_PORT=port)
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_gloo_world2_bucketed_gradient_all_reduce():
    """Training row N3 on CPU: ``grad_sync.BucketedAllReduce`` over two gloo ranks -- gradients live in contiguous buckets,
    reductions are launched from the backward hooks, ``finish()`` averages and sanitises; the result must equal the
    gradient of the mean loss over both ranks' data, with several buckets, an unused parameter, a NaN to be sanitised and a
    second step after ``zero_grad()`` (and after an optimiser that drops the gradients)."""
    script = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd.grad_sync import BucketedAllReduce
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 48), torch.nn.ReLU(), torch.nn.Linear(48, 1))
unused = torch.nn.Parameter(torch.ones(5))
params = list(net.parameters()) + [unused]
sync = BucketedAllReduce(params, bucket_bytes=8192)          # 2048 floats per bucket -> several buckets
assert len(sync.buckets) >= 3
data = [torch.randn(16, 32, generator=torch.Generator().manual_seed(10 + k)) for k in range(2)]
def ref_grads(scale):
    gs = []
    for k in range(2):
        net2 = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 48), torch.nn.ReLU(), torch.nn.Linear(48, 1))
        net2.load_state_dict(net.state_dict())
        (net2(data[k]).square().mean() * scale).backward()
        gs.append([p.grad.clone() for p in net2.parameters()])
    return [(a + b) / 2 for a, b in zip(*gs)]
for step, scale in enumerate((1.0, 3.0)):
    want = ref_grads(scale)
    if step == 0:
        sync.zero_grad()
    else:
        for p in params:                 # what optimizer.zero_grad(set_to_none=True) does
            p.grad = None
        sync.zero_grad()
    if step == 0:
        sync.arm()                       # reductions launched from the hooks; step 1 stays un-armed: everything reduced in finish()
    (net(data[r]).square().mean() * scale).backward()
    assert (len(sync._handles) > 0) == (step == 0)
    if step == 0 and r == 1:
        sync.buckets[0][0] = float("nan")            # a non-finite value on one rank must not survive
    sync.finish()
    for p, w in zip(net.parameters(), want):
        if step == 0 and p.grad.data_ptr() == sync.buckets[0].data_ptr():
            assert torch.isfinite(p.grad).all()
            assert torch.allclose(p.grad.reshape(-1)[1:], w.reshape(-1)[1:], rtol=1e-5, atol=1e-7)
        else:
            assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-7), (step, p.shape)
    assert unused.grad is not None and float(unused.grad.abs().max()) == 0.0
    # gradients are views into the buckets (one fused optimiser / sanitiser pass per bucket)
    assert all(any(p.grad.data_ptr() >= b.data_ptr() and p.grad.data_ptr() < b.data_ptr() + 4 * b.numel() for b in sync.buckets) for p in params)
sync.remove()
dist.destroy_process_group()
print("rank", r, "ok")
'''
    port = str(33500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), SHG_ROOT=ROOT, SHG_PORT=port)
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_gloo_world2_two_backward_passes_per_phase_reduce_once():
    """one phase runs several backward passes into the same buckets (Dgen then Dreal,
    stylegan_default_loss.py:93,126; effective_batch_gpu rounds).  Only the armed, LAST pass may launch reductions; the result
    is the world mean of the SUM of every pass's gradient.  The advisor's repro: per-rank gradients (1, 10) then (2, 20) -> 16.5."""
    script = r"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd.grad_sync import BucketedAllReduce
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"], rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
w = torch.nn.Parameter(torch.zeros(3)); v = torch.nn.Parameter(torch.zeros(2000)); u = torch.nn.Parameter(torch.zeros(4))
sync = BucketedAllReduce([w, v, u], bucket_bytes=4096)
assert len(sync.buckets) >= 2
for trial in range(3):
    sync.zero_grad()
    first, second = (1.0, 2.0) if r == 0 else (10.0, 20.0)
    ((w.sum() + v.sum()) * first).backward()             # un-armed: accumulate only (u not reached at all)
    assert not sync._handles
    time.sleep(0.05)
    ((w.sum() + v.sum()) * first).backward()             # a second accumulation round, still un-armed
    sync.arm()
    ((w.sum() + v.sum() + (u.sum() if trial == 1 else 0.0)) * second).backward()
    sync.finish()
    want = (2 * 1.0 + 2.0 + 2 * 10.0 + 20.0) / 2
    assert torch.allclose(w.grad, torch.full_like(w, want)) and torch.allclose(v.grad, torch.full_like(v, want)), (w.grad, want)
    assert torch.allclose(u.grad, torch.full_like(u, 11.0 if trial == 1 else 0.0))
    assert [id(p) for p in sync.untouched()] == ([] if trial == 1 else [id(u)])
try:
    sync.arm(); sync.arm()
    raise SystemExit("double arm() must raise")
except RuntimeError:
    pass
dist.destroy_process_group()
print("rank", r, "ok")
"""
    port = str(37500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), SHG_ROOT=ROOT, SHG_PORT=port)
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_gloo_world2_deferred_reduction_of_the_split_graph_form():
    """``train_stage.PhaseGraphs`` with more than one rank replays the backward passes as a HIP graph: no hook runs on the host, so the
    buckets are reduced afterwards -- ``defer`` keeps the armed hooks from launching a collective (what a capture needs), ``reduce_all()``
    puts every bucket in flight, ``finish(reduced=True)`` only averages and sanitises.  Same result as the hook-launched form, and a
    NaN / inf in one rank's gradient is sanitised after the reduction exactly as there."""
    script = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd.grad_sync import BucketedAllReduce
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"], rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
w = torch.nn.Parameter(torch.zeros(3)); v = torch.nn.Parameter(torch.zeros(2000)); u = torch.nn.Parameter(torch.zeros(4))
sync = BucketedAllReduce([w, v, u], bucket_bytes=4096)
assert len(sync.buckets) >= 2
out = {}
for form in ("hooks", "deferred"):
    sync.zero_grad()
    sync.defer = form == "deferred"
    a, b = (1.0, 2.0) if r == 0 else (10.0, 20.0)
    ((w.sum() + v.sum()) * a).backward()
    sync.arm()
    ((w.sum() + v.sum()) * b + (v[:1].sum() * float("inf") if r == 1 else 0.0)).backward()
    if form == "deferred":
        assert not sync._handles                         # the armed pass launched nothing
        sync.defer = False
        sync.reduce_all()
        sync.finish(reduced=True)
    else:
        assert sync._handles
        sync.finish()
    out[form] = torch.cat([w.grad.reshape(-1), v.grad.reshape(-1), u.grad.reshape(-1)]).clone()
    assert [id(p) for p in sync.untouched()] == [id(u)]
assert torch.equal(out["hooks"], out["deferred"]), (out["hooks"][:5], out["deferred"][:5])
want = (1.0 + 2.0 + 10.0 + 20.0) / 2
assert torch.allclose(out["hooks"][:3], torch.full((3,), want)) and float(out["hooks"][3]) == 1e5      # v[0]: inf -> nan_to_num(posinf=1e5)
dist.destroy_process_group()
print("rank", r, "ok")
"""
    port = str(39500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), SHG_ROOT=ROOT, SHG_PORT=port)
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_stream_pipeline_is_a_plain_loop_without_a_gpu():
    import torch
    from shgan_amd import eval_harness as hz
    pipe = hz.StreamPipeline('cpu')
    assert pipe.streams == []
    outs = [pipe.run(lambda a, b: a + b, torch.full((3,), float(k)), torch.ones(3)) for k in range(4)]
    pipe.join()
    assert [float(o[0]) for o in outs] == [1.0, 2.0, 3.0, 4.0]
    assert hz.PIPELINE_DEPTH >= 1
