"""Eval-loop caller of the generator: the body of
lib/experiments/shgan_default.py:257-289 -- input assembly, generator call, uint8 composite -- and the
batch-sharded multi-GPU loop (one process per GPU, rank-strided samples, no collective on the data
path; an optional RCCL all-gather returns the uint8 results in dataset order)."""
import numpy as np
import torch

from . import kernels
from .data import DistributedSampler, RandomMask, zipzap_arrange


def assemble_input(real, mask):
    """real [N,3,R,R] in [-1,1], mask [N,1,R,R] in {0,1} -> x = cat([mask-0.5, real*mask]) (shgan_default.py:270-274)."""
    if real.is_cuda:
        return kernels.assemble_input(real.float(), mask.float())       # one HIP kernel instead of sub + mul + cat
    return torch.cat([mask - 0.5, real * mask], dim=1)                  # host tensors (dataloader side): plain torch


def run_generator(G, x, z, c=None, noise_mode='random'):
    """x [N,4,R,R], z [N,z_dim] -> uint8 [N,3,R,R]: generated pixels inside the hole, the known pixels
    elsewhere (shgan_default.py:257-262; float->uint8 is a truncation)."""
    if c is None:
        c = torch.zeros([x.shape[0], G.c_dim], device=x.device)
    with torch.no_grad():
        img = G(x=x, z=z, c=c, noise_mode=noise_mode)
    return kernels.composite_u8(x, img)


def synthetic_batch(n, resolution, z_dim=512, seed=0, device='cuda', masks='freeform'):
    """Synthetic masked inputs of SURVEY.md 8(d): real ~ U{0..255}/127.5-1; masks from RandomMask seeded
    with ``seed`` ('freeform') or Bernoulli(0.7) per pixel ('bernoulli', cheap for throughput runs)."""
    rs = np.random.RandomState(seed)
    real_u8 = rs.randint(0, 256, size=(n, 3, resolution, resolution)).astype(np.uint8)
    if masks == 'freeform':
        np.random.seed(seed)
        mask = np.stack([RandomMask(resolution, [0, 1]) for _ in range(n)]).astype(np.uint8)
    else:
        mask = (rs.rand(n, 1, resolution, resolution) < 0.7).astype(np.uint8)
    z = torch.from_numpy(rs.standard_normal((n, z_dim)).astype(np.float32))
    real = torch.from_numpy(real_u8.astype(np.float32)) / 127.5 - 1.0
    m = torch.from_numpy(mask.astype(np.float32))
    return assemble_input(real, m).to(device), z.to(device), real_u8, mask


def sharded_eval(G, n_items, batch_size, resolution, rank=0, world=1, seed=0, gather=True, device='cuda',
                 noise_mode='const', masks='bernoulli'):
    """Batch-sharded evaluation over a synthetic dataset of ``n_items`` images: rank r processes the
    sample ids ``DistributedSampler(extend=True)`` gives it, in batches of ``batch_size``; with
    ``gather`` the uint8 outputs are all-gathered (RCCL) and re-interleaved to dataset order.
    Item i's inputs depend only on (seed, i), so any world size produces the same per-item results."""
    import torch.distributed as dist
    ids = list(iter(DistributedSampler(list(range(n_items)), num_replicas=world, rank=rank, shuffle=False, extend=True)))
    outs = []
    for b0 in range(0, len(ids), batch_size):
        chunk = ids[b0:b0 + batch_size]
        xs, zs = [], []
        for i in chunk:
            x, z, _, _ = synthetic_batch(1, resolution, G.z_dim, seed=seed * 1000003 + i, device=device, masks=masks)
            xs.append(x)
            zs.append(z)
        outs.append(run_generator(G, torch.cat(xs), torch.cat(zs), noise_mode=noise_mode))
    local = torch.cat(outs)
    if not gather or world == 1:
        return ids, local
    full = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local)
    per_rank_ids = [list(iter(DistributedSampler(list(range(n_items)), num_replicas=world, rank=r, shuffle=False, extend=True)))
                    for r in range(world)]
    order = zipzap_arrange(per_rank_ids)[:n_items]
    merged = zipzap_arrange([full[r].cpu().numpy() for r in range(world)])[:n_items]
    return order, merged
