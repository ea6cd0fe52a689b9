"""Eval-loop caller of the generator: the body of
lib/experiments/shgan_default.py:257-289 -- input assembly, generator call, uint8 composite -- and the
batch-sharded multi-GPU loop (one process per GPU, rank-strided samples, no collective on the data
path; an optional all-gather returns the uint8 results in dataset order: RCCL over xGMI on GPUs,
gloo on the CPU for the world-size-2 tests)."""
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
    if masks == 'freeform' and torch.device(device).type == 'cuda' and resolution % 32 == 0 and resolution <= 512:
        # same masks bit for bit, drawn on the device (sh-gan_amd/masks.py): host RNG draws + HIP rasteriser, no H2D of pixels
        from . import masks as dev_masks
        np.random.seed(seed)
        m = dev_masks.random_masks(n, resolution, [0, 1], device=device)
        z = torch.from_numpy(rs.standard_normal((n, z_dim)).astype(np.float32))
        real = (torch.from_numpy(real_u8.astype(np.float32)) / 127.5 - 1.0).to(device)
        return assemble_input(real, m), z.to(device), real_u8, m.cpu().numpy().astype(np.uint8)
    if masks == 'freeform':
        np.random.seed(seed)
        mask = np.stack([RandomMask(resolution, [0, 1]) for _ in range(n)]).astype(np.uint8)
    else:
        mask = (rs.rand(n, 1, resolution, resolution) < 0.7).astype(np.uint8)
    z = torch.from_numpy(rs.standard_normal((n, z_dim)).astype(np.float32))
    real = torch.from_numpy(real_u8.astype(np.float32)) / 127.5 - 1.0
    m = torch.from_numpy(mask.astype(np.float32))
    return assemble_input(real, m).to(device), z.to(device), real_u8, mask


def synthetic_items(ids, resolution, z_dim=512, seed=0, device='cuda'):
    """Inputs of the dataset items ``ids`` of a synthetic eval set, built ON ``device`` in one batch (no per-image host
    work or H2D copy): item i is drawn from its own generator seeded with (seed, i), so it is the same image whichever
    rank / batch position processes it.  -> x [B,4,R,R], z [B,z_dim], real_u8 [B,3,R,R], mask [B,1,R,R] (all on device)."""
    dev = torch.device(device)
    b = len(ids)
    real_u8 = torch.empty((b, 3, resolution, resolution), dtype=torch.uint8, device=dev)
    mask = torch.empty((b, 1, resolution, resolution), dtype=torch.float32, device=dev)
    z = torch.empty((b, z_dim), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev)
    for k, i in enumerate(ids):
        g.manual_seed(int(seed) * 1000003 + int(i))
        real_u8[k].random_(0, 256, generator=g)
        mask[k].bernoulli_(0.7, generator=g)
        z[k].normal_(generator=g)
    real = real_u8.to(torch.float32).div_(127.5).sub_(1.0)
    return assemble_input(real, mask), z, real_u8, mask


def shard_ids(n_items, rank, world):
    """Sample ids of rank ``rank``: ``DistributedSampler(shuffle=False, extend=True)`` = indices[rank::world] after the
    extend-pad with leading indices (ds_sampler.py:58-68)."""
    return list(iter(DistributedSampler(list(range(n_items)), num_replicas=world, rank=rank, shuffle=False, extend=True)))


PIPELINE_DEPTH = 3      # streams of the evaluation loop (1 = plain loop); 2: +6.5 %, 3: +8.5 %, 4-6: no further gain at 512x16


class StreamPipeline:
    """Round-robin HIP streams for an evaluation loop.  Consecutive batches are independent, and a forward pass is ~150 kernel
    launches back to back on one stream: every launch boundary idles the chip for a few microseconds (the last workgroups of a
    kernel drain, the next kernel's first ones ramp up).  With batch k+1 queued on a second stream the other batch's kernels
    fill those gaps -- 759 -> 807 (two streams) -> 824 images/s (three) at 512x16 on one MI355X.  ``depth`` = 1 is the plain loop.

        pipe = StreamPipeline(device, depth=2)
        for x, z in batches: outs.append(pipe.run(step_fn, x, z))
        pipe.join()                      # before anything reads ``outs`` on the current stream"""

    def __init__(self, device, depth=None):
        depth = PIPELINE_DEPTH if depth is None else depth
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(depth)] if (depth > 1 and self.device.type == 'cuda') else []
        self.k = 0
        self.outs = []

    def run(self, fn, *args):
        if not self.streams:
            return fn(*args)
        if self.k == 0:
            # the first batch runs on the caller's stream: it builds the per-parameter caches (prepared weight layouts, separable
            # filter taps, ...) in stream order; every later batch waits for the caller's stream before it starts, i.e. for them
            self.k = 1
            out = fn(*args)
            self.outs.append(out)
            return out
        s = self.streams[self.k % len(self.streams)]
        self.k += 1
        s.wait_stream(torch.cuda.current_stream(self.device))      # the inputs were produced on the caller's stream
        with torch.cuda.stream(s):
            out = fn(*args)
        for t in args:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(s)
        self.outs.append(out)
        return out

    def join(self):
        cur = torch.cuda.current_stream(self.device) if self.streams else None
        for s in self.streams:
            cur.wait_stream(s)
        for o in self.outs:
            if torch.is_tensor(o) and o.is_cuda:
                o.record_stream(cur)
        self.outs = []


class GraphPipeline:
    """The same round-robin loop with every slot's forward pass captured once as a HIP graph (``torch.cuda.graph``: the ~150 ctypes
    launches, the allocator traffic and the device RNG of ``fn`` become one ``hipGraphLaunch`` per batch).  On one GPU the host
    keeps up with eager launches anyway (2 ms of Python per 19 ms step); with eight ranks on one host eight interpreters compete
    for cores, and the replay takes the host out of the step.  ``fn(*args)`` must be shape-static and sync-free; per-parameter
    caches are warmed by eager calls before the capture.  ``run(*args)`` copies the arguments into the slot's static inputs and
    replays; the returned tensor is the slot's static output (valid until that slot runs again, ``depth`` calls later).

        pipe = GraphPipeline(device, step_fn, (x0, z0), depth=3)
        for x, z in batches: out = pipe.run(x, z); ...
        pipe.join()"""

    def __init__(self, device, fn, example_args, depth=None, warmup=2, watch=None):
        """``watch``: the parameters / buffers ``fn`` reads (e.g. ``list(G.parameters())``).  A captured graph bakes in the ADDRESSES of
        the per-parameter caches of the modules (prepared weight layouts, packed fp16 weights, the host-read noise strength): when a
        parameter changes -- optimiser step, EMA update, ``load_state_dict`` (version counters) or a replayed training graph
        (``_ParamCache.invalidate_all()``: the epoch) -- the next eager call rebuilds those caches and frees the tensors the graph still
        points at.  With ``watch`` the pipeline compares that signature before every replay and re-captures all slots when it moved;
        without it the parameters must stay frozen for the pipeline's lifetime."""
        depth = PIPELINE_DEPTH if depth is None else max(1, depth)
        self.device = torch.device(device)
        self.fn, self.warmup = fn, max(1, warmup)
        self.watch = None if watch is None else list(watch)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(depth)]
        self.ins = [[a.clone() if torch.is_tensor(a) else a for a in example_args] for _ in self.streams]
        self.k = 0
        self.captures = 0
        self._capture()

    def _signature(self):
        if self.watch is None:
            return None
        from .model_zoo.stylegan import _ParamCache
        return (_ParamCache.epoch,) + tuple((t.data_ptr(), t._version) for t in self.watch)

    def _capture(self):
        cur = torch.cuda.current_stream(self.device)
        self.graphs, self.outs = [], []            # (drops the previous graphs and their private pools)
        for s, ins in zip(self.streams, self.ins):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(self.warmup):       # eager: rebuilds the per-parameter caches from the live parameters
                    self.fn(*ins)
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = self.fn(*ins)
            self.graphs.append(g)
            self.outs.append(out)
        cur.wait_stream(self.streams[-1])
        self.sig = self._signature()
        self.captures += 1

    def run(self, *args):
        if self.watch is not None and self._signature() != self.sig:
            self.join()
            torch.cuda.current_stream(self.device).synchronize()
            self._capture()
        i = self.k % len(self.streams)
        self.k += 1
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for dst, a in zip(self.ins[i], args):
                if torch.is_tensor(dst) and dst.data_ptr() != a.data_ptr():
                    dst.copy_(a, non_blocking=True)
            self.graphs[i].replay()
        return self.outs[i]

    def join(self):
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)


def sharded_eval(G, n_items, batch_size, resolution, rank=0, world=1, seed=0, gather=True, device='cuda',
                 noise_mode='const', step_fn=None, z_dim=None, pipeline_depth=None):
    """Batch-sharded evaluation over a synthetic dataset of ``n_items`` images (BASELINE config 4): rank r processes
    the sample ids ``shard_ids`` gives it, in batches of ``batch_size`` built on the device; with ``gather`` the uint8
    outputs are all-gathered (one ``all_gather_into_tensor`` per run: RCCL on GPUs, gloo on CPU tensors) and
    re-interleaved to dataset order (eva_base.py:196-230).  ``step_fn(x, z) -> uint8 [B,3,R,R]`` replaces the generator
    step (the CPU world-size-2 test injects a stand-in; the product path is ``run_generator(G, ...)``).
    With ``noise_mode='random'`` the torch generator is seeded per rank as the reference does (seed*world + rank,
    shgan_default.py:165-167)."""
    import torch.distributed as dist
    if step_fn is None:
        def step_fn(x, z):
            return run_generator(G, x, z, noise_mode=noise_mode)
    if z_dim is None:
        z_dim = G.z_dim
    if noise_mode == 'random':
        torch.manual_seed(seed * world + rank)
    ids = shard_ids(n_items, rank, world)
    outs = []
    pipe = StreamPipeline(device, depth=pipeline_depth)
    for b0 in range(0, len(ids), batch_size):
        x, z, _, _ = synthetic_items(ids[b0:b0 + batch_size], resolution, z_dim, seed=seed, device=device)
        outs.append(pipe.run(step_fn, x, z))
    pipe.join()
    local = torch.cat(outs)
    if not gather or (world == 1 and not (dist.is_available() and dist.is_initialized())):
        return ids, local          # (a 1-rank process group still goes through the collective: same code path as N ranks)
    # concatenated-along-dim-0 form: the one layout both RCCL and gloo accept for all_gather_into_tensor
    full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous())
    full = full.view((world,) + tuple(local.shape))
    per_rank_ids = [shard_ids(n_items, r, world) for r in range(world)]
    order = zipzap_arrange(per_rank_ids)[:n_items]
    merged = zipzap_arrange([full[r].cpu().numpy() for r in range(world)])[:n_items]
    return order, merged
