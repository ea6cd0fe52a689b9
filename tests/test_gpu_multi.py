"""RCCL with N > 1 ranks -- runs by itself wherever ``torch.cuda.device_count() >= 2`` (the 1-GPU test boxes skip it; the CPU suite covers
the same code over gloo, world 2).  Two entry points: ``bench.py --gpus 2`` (its own launcher, one rank per GPU, backend nccl) and
``eval_harness.sharded_eval(gather=True)`` + ``grad_sync.BucketedAllReduce`` in a 2-rank nccl group (batch sharding of config 4 and the
gradient all-reduce of config 5)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
needs2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs (RCCL with more than one rank)')


@needs2
def test_bench_two_ranks_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                        '--no-second-config', '--all-blocks', '--train-steps', '1'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    line = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['collective_backend'] == 'nccl' and line['config']['ranks_all_reduced'] == 2
    assert not line['config']['ranks_share_devices'] and line['config']['global_batch'] == 32
    assert line['train_step']['grad_all_reduce'] == 'nccl' and line['train_step']['losses_finite']


@needs2
def test_sharded_eval_and_gradient_buckets_two_ranks_over_rccl():
    script = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd import configs, eval_harness as hz
from shgan_amd.grad_sync import BucketedAllReduce
r = int(os.environ["RANK"])
torch.cuda.set_device(r)
dev = torch.device("cuda", r)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"], rank=r, world_size=2, device_id=dev)
kw = dict(ch_base=2048, ch_max=32, w_dim=64, z_dim=48, w0_dim=96)
G = configs.seeded_init_(configs.build_generator(256, **kw), seed=5).eval().requires_grad_(False).to(dev)
with torch.no_grad():
    order, merged = hz.sharded_eval(G, n_items=7, batch_size=2, resolution=256, seed=3, device=dev, rank=r, world=2, gather=True, z_dim=48)
    ids, alone = hz.sharded_eval(G, n_items=7, batch_size=2, resolution=256, seed=3, device=dev, rank=0, world=1, gather=False, z_dim=48)
assert order == list(range(7)) and np.array_equal(merged, alone.cpu().numpy())      # sharded over two GPUs == one GPU, in dataset order
w = torch.nn.Parameter(torch.zeros(5000, device=dev))
sync = BucketedAllReduce([w], bucket_bytes=8192)
sync.zero_grad()
(w.sum() * (1.0 if r == 0 else 10.0)).backward()
sync.arm()
(w.sum() * (2.0 if r == 0 else 20.0)).backward()
sync.finish()
assert torch.allclose(w.grad, torch.full_like(w, 16.5))
dist.destroy_process_group()
print("rank", r, "ok")
'''
    port = str(36500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), SHG_ROOT=ROOT, SHG_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0 and f'rank {rank} ok'.encode() in out, out.decode()[-3000:]
