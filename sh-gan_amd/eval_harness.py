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
        # one HIP kernel instead of sub + mul + cat; decoded uint8 pixels are converted in the same pass (kernels.u8_value_table)
        return kernels.assemble_input(real if real.dtype == torch.uint8 else real.float(), mask.float())
    if real.dtype == torch.uint8:                                       # decoded pixels: ToTensor (/255) + the formatter's *2-1
        real = kernels.u8_value_table('cpu')[real.long()]
    return torch.cat([mask - 0.5, real * mask], dim=1)                  # host tensors (dataloader side): plain torch


def run_generator(G, x, z, c=None, noise_mode='random', out=None):
    """x [N,4,R,R], z [N,z_dim] -> uint8 [N,3,R,R]: generated pixels inside the hole, the known pixels
    elsewhere (shgan_default.py:257-262; float->uint8 is a truncation).  ``out``: write into this uint8 tensor (a slice of
    the evaluation loop's result buffer)."""
    if c is None:
        c = torch.zeros([x.shape[0], G.c_dim], device=x.device)
    with torch.no_grad():
        img = G(x=x, z=z, c=c, noise_mode=noise_mode)
    return kernels.composite_u8(x, img, out=out)


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


_STREAMS = {}


def shared_streams(device, n, kind='pipe'):
    """The first ``n`` of this process's ``kind`` streams on ``device`` (created on first use, then reused by every pipeline / feeder).
    HIP streams are multiplexed onto a handful of hardware queues (four by default): every ``torch.cuda.Stream()`` takes the next of
    torch's pooled streams, and a loop that builds fresh streams each time it starts ends up -- after a few pipelines have come and
    gone -- with its staging stream on the hardware queue of one of its generator streams, where a 12 MB copy or a mask raster waits
    behind 19 ms of convolutions (measured: the evaluation loop 20.5 -> 44 ms per batch inside bench.py, round 6).  One small fixed
    set per process keeps the caller's stream, the copy stream and the three generator streams on queues of their own."""
    dev = torch.device(device)
    key = (str(dev), kind)
    have = _STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]


PIPELINE_DEPTH = 3      # streams of the evaluation loop (1 = plain loop); 2: +6.5 %, 3: +8.5 %, 4-6: no further gain at 512x16


class StreamPipeline:
    """Round-robin HIP streams for an evaluation loop.  Consecutive batches are independent, and a forward pass is ~150 kernel
    launches back to back on one stream: every launch boundary idles the chip for a few microseconds (the last workgroups of a
    kernel drain, the next kernel's first ones ramp up).  With batch k+1 queued on a second stream the other batch's kernels
    fill those gaps -- 759 -> 807 (two streams) -> 824 images/s (three) at 512x16 on one MI355X.  ``depth`` = 1 is the plain loop.

        pipe = StreamPipeline(device, depth=2)
        for x, z in batches: outs.append(pipe.run(step_fn, x, z))
        pipe.join()                      # before anything reads ``outs`` on the current stream"""

    def __init__(self, device, depth=None, first_on_caller=True):
        """``first_on_caller=False``: the first batch goes to a side stream too and every side stream's first batch waits for an event
        behind it (the per-parameter caches it built) -- for loops whose caller's stream must stay short because the host waits on
        it (EvalLoop: the mask rasteriser's hole-count read)."""
        depth = PIPELINE_DEPTH if depth is None else depth
        self.device = torch.device(device)
        self.streams = shared_streams(self.device, depth) if (depth > 1 and self.device.type == 'cuda') else []
        self.k = 0
        self.outs = []
        self.last_stream = None        # the stream the latest batch was issued on (None: the caller's)
        self.first_on_caller = first_on_caller
        self._first_done = None

    def run(self, fn, *args):
        self.last_stream = None
        if not self.streams:
            return fn(*args)
        if self.k == 0 and not self.first_on_caller:
            s = self.streams[0]
            self.k = 1
            self.last_stream = s
            s.wait_stream(torch.cuda.current_stream(self.device))
            from .model_zoo.stylegan import _ParamCache
            built = _ParamCache.builds
            with torch.cuda.stream(s):
                out = fn(*args)
                if _ParamCache.builds != built:        # the batch prepared weights on this stream: the other streams' first batches wait for it
                    self._first_done = torch.cuda.Event()
                    self._first_done.record(s)
            for t in args:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(s)
            self.outs.append(out)
            return out
        if self.k == 0:
            # the first batch runs on the caller's stream: it builds the per-parameter caches (prepared weight layouts, separable
            # filter taps, ...) in stream order; every later batch waits for the caller's stream before it starts, i.e. for them
            self.k = 1
            out = fn(*args)
            self.outs.append(out)
            return out
        s = self.streams[self.k % len(self.streams)]
        self.k += 1
        self.last_stream = s
        s.wait_stream(torch.cuda.current_stream(self.device))      # the inputs were produced on the caller's stream
        if self._first_done is not None and self.k <= len(self.streams) + 1:
            s.wait_event(self._first_done)                         # the caches the first batch built on its side stream
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
            for t in (o if isinstance(o, (tuple, list)) else (o,)):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
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
        self.streams = shared_streams(self.device, depth)
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


def zipzap_device(full, n_items):
    """``zipzap_arrange`` (eva_base.py:196-230) of equally long rank shards, on the device: full [world, n_local, ...] (rank r's
    k-th result at [r, k]) -> [n_items, ...] in dataset order (item k*world + r), the padded duplicates of
    ``DistributedSampler(extend=True)`` cut off.  One transposing copy instead of a host round trip of the whole result set."""
    world, n_local = full.shape[:2]
    return full.transpose(0, 1).reshape((world * n_local,) + tuple(full.shape[2:]))[:n_items]


def standin_features(images_u8, dim=2048):
    """Stand-in for the feature detector of the FID stage (eva_fid.py:145-158,194-206: an Inception-v3 TorchScript download, not
    reproducible offline): a fixed linear map of the uint8 images to [B, dim] -- the mean of ``dim`` contiguous pixel runs.  It only
    gives the moment accumulation, the collectives and the loop's timing a feature tensor of the real shape; it measures nothing."""
    b = images_u8.shape[0]
    flat = images_u8.reshape(b, -1)
    per = flat.shape[1] // dim
    return flat[:, :per * dim].reshape(b, dim, per).to(torch.float32).mean(dim=2)


def broadcast_state(module, src=0, group=None):
    """Rank ``src``'s parameters and buffers to every rank in ONE flat float32 ``dist.broadcast`` (RCCL over xGMI on GPUs, gloo on
    CPU tensors): what the reference gets from the ``DistributedDataParallel`` constructor after rank 0 alone has read the
    checkpoint (shgan_default.py:138-154,223-231; 317 MB for the 512 generator, SURVEY 8(e) collective (1)).  Non-float32 state
    (integer counters) travels in a second, small broadcast.  Parameters are written with ``copy_`` so the prepared-weight caches
    (version counters) follow.  No-op without an initialised process group; a 1-rank group runs the collective too.
    Returns the number of bytes broadcast."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    tensors = [t for _, t in sorted(list(module.named_parameters()) + list(module.named_buffers()), key=lambda kv: kv[0])]
    f32 = [t for t in tensors if t.dtype == torch.float32]
    other = [t for t in tensors if t.dtype != torch.float32]
    total = 0
    with torch.no_grad():
        for grp, dt in ((f32, torch.float32), (other, torch.float64)):
            if not grp:
                continue
            dev = grp[0].device
            flat = torch.empty(sum(t.numel() for t in grp), dtype=dt, device=dev)
            if dist.get_rank(group) == src:
                off = 0
                for t in grp:
                    flat[off:off + t.numel()].copy_(t.detach().reshape(-1))
                    off += t.numel()
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in grp:
                t.copy_(flat[off:off + t.numel()].reshape(t.shape).to(t.dtype))
                off += t.numel()
            total += flat.numel() * flat.element_size()
    return total


class EvalLoop:
    """The evaluation loop the path exists for (lib/experiments/shgan_default.py:264-300; BASELINE config 4 = one rank of it), as a
    streamed loop on ONE rank's shard:

        per batch   loader (host: decoded uint8 or float images, ids)  ->  H2D on the copy stream under the previous batch's kernels
                    (datasets.DeviceFeeder)  ->  freeform masks drawn on the device (masks.random_masks, the reference's numpy
                    draws in its order)  ->  x = cat([mask - .5, real * mask])  ->  z ~ N(0, 1)  ->  G + uint8 composite written
                    INTO the rank's result buffer at the batch's position  ->  features (``feature_fn``, the detector hand-off of
                    eva_fid.py:194-206)  ->  fp64 moments on the device (fid_stats.FidStats; padded duplicates weigh 0);
        at the end  ONE all-reduce of the moments and ONE all-gather of the uint8 results + the zipzap re-interleave on the device
                    (the reference: 3 x world broadcasts per batch, eva_base.py:96-188, python lists on rank 0).

    Consecutive batches are issued round-robin on ``depth`` HIP streams (StreamPipeline, every batch on a side stream); the moment
    kernel accumulates in place, so every stream owns a partial accumulator (33.6 MB of float64 each) and ``gather`` adds them up.
    The only point where the host waits for the device is the mask rasteriser's hole-count read (one small D2H per batch), and it
    waits on the COPY stream, which carries nothing but input staging.  The loop uses five streams -- caller's, copy, three for the
    generator; a sixth (a statistics stream, the first form of this loop) made the staging stream share a hardware queue with a
    generator stream and the hole-count read waited for a whole batch (MEASUREMENTS.md, round 6).  ``latent_fn(ids, B) -> z``
    replaces ``torch.randn`` (tests: per-item latents so that a result can be compared id by id); ``on_batch(ids, images_u8, event)``
    hands every finished batch to a consumer (host metrics) without ending the loop."""

    def __init__(self, G, device, resolution, n_items, rank=0, world=1, noise_mode='random', seed=0, depth=None, feature_fn=None,
                 fid_dim=2048, latent_fn=None, device_masks=True, hole_range=(0, 1), keep_images=True, on_batch=None, step_fn=None,
                 fid_accumulate_fn=None, feeder_stream=False, timing=False):
        from .datasets import DeviceFeeder
        self.timing, self.batch_done_events = timing, []      # timing: one timing event per finished batch (bench: steady-state rate)
        self.G, self.device, self.res = G, torch.device(device), int(resolution)
        self.n_items, self.rank, self.world = int(n_items), int(rank), int(world)
        self.noise_mode, self.seed, self.depth = noise_mode, seed, depth
        self.feature_fn, self.latent_fn, self.on_batch = feature_fn, latent_fn, on_batch
        self.step_fn = step_fn          # (x4, z, out) -> uint8 images: the CPU world-size-2 tests inject a stand-in; the product path is run_generator
        self.ids = shard_ids(self.n_items, self.rank, self.world)
        self.feeder = DeviceFeeder(self.device, self.res, hole_range=hole_range, device_masks=device_masks, own_stream=feeder_stream)
        self.fid_dim, self._fid_fn = fid_dim, fid_accumulate_fn
        self._fid_parts = {}            # stream id -> FidStats (partial sums of the batches that ran on that stream)
        self.fid = None                 # their sum, after gather()
        self.images = (torch.empty((len(self.ids), 3, self.res, self.res), dtype=torch.uint8, device=self.device) if keep_images else None)
        self.seen = 0

    def _fid_part(self, key):
        from .fid_stats import FidStats
        if key not in self._fid_parts:
            self._fid_parts[key] = FidStats(self.fid_dim, device=self.device, accumulate_fn=self._fid_fn)
        return self._fid_parts[key]

    def run(self, loader):
        """``loader`` yields this rank's items in ``shard_ids`` order as (images [B,3,R,R] uint8 or float32 in [-1,1], ids) or
        (images, masks [B,R,R], ids).  Returns self (``images``, ``seen``; ``gather`` completes ``fid``)."""
        if self.noise_mode == 'random':
            torch.manual_seed(self.seed * self.world + self.rank)          # shgan_default.py:165-167
        pipe = StreamPipeline(self.device, depth=self.depth, first_on_caller=False)
        G, buf = self.G, self.images
        if self.feature_fn is not None:                                    # accumulators exist before a side stream touches them
            for key in [None] + [st.cuda_stream for st in pipe.streams]:
                self._fid_part(key)
        for x4, real, mask, ids in self.feeder(loader):
            b, k0 = x4.shape[0], self.seen
            if k0 + b > len(self.ids):
                raise ValueError(f'EvalLoop: the loader yielded more than the {len(self.ids)} items of this rank\'s shard')
            z = self.latent_fn(ids, b) if self.latent_fn is not None else torch.randn([b, G.z_dim], device=self.device)
            gen = self.step_fn if self.step_fn is not None else (lambda x_, z_, o_: run_generator(G, x_, z_, noise_mode=self.noise_mode, out=o_))
            dst = buf[k0:k0 + b] if buf is not None else None

            def step(x4_, z_, dst=dst, k0=k0):
                out = gen(x4_, z_, dst)
                if self.feature_fn is not None:
                    cur = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == 'cuda' else None
                    part = self._fid_parts.get(cur) or self._fid_part(None)
                    part.add_shard(self.feature_fn(out), k0, self.rank, self.world, self.n_items)
                return out
            out = pipe.run(step, x4, z)
            if self.timing and self.device.type == 'cuda':
                tev = torch.cuda.Event(enable_timing=True)
                tev.record(pipe.last_stream or torch.cuda.current_stream(self.device))
                self.batch_done_events.append(tev)
            if self.on_batch is not None:
                ev = None
                if self.device.type == 'cuda':
                    ev = torch.cuda.Event()
                    ev.record(pipe.last_stream or torch.cuda.current_stream(self.device))
                self.on_batch(ids, out, ev)
            self.seen += b
        pipe.join()
        return self

    def gather(self):
        """-> (uint8 images [n_items,3,R,R] in dataset order on the device, FidStats summed over the ranks | None).  One
        ``all_gather_into_tensor`` + ``zipzap_device``; one ``all_reduce`` of the moments.  Every rank must have run its whole shard."""
        import torch.distributed as dist
        if self.seen != len(self.ids):
            raise ValueError(f'EvalLoop.gather: {self.seen} of {len(self.ids)} items of this rank\'s shard were processed')
        use = dist.is_available() and dist.is_initialized()
        images = None
        if self.images is not None:
            if use:
                # RCCL gathers device tensors over xGMI; a gloo group (CPU tests, ranks sharing one device) is handed host tensors
                via_host = dist.get_backend() == 'gloo' and self.images.is_cuda
                local = self.images.cpu() if via_host else self.images
                full = torch.empty((self.world,) + tuple(local.shape), dtype=torch.uint8, device=local.device)
                dist.all_gather_into_tensor(full.view((-1,) + tuple(local.shape[1:])), local)
                full = full.to(self.device)
            else:
                full = self.images[None]
            images = zipzap_device(full, self.n_items)
        if self.local_fid() is not None:
            self.fid.all_reduce()
        return images, self.fid

    def local_fid(self):
        """This rank's moments (the per-stream partial accumulators added up; no collective)."""
        if self._fid_parts:
            parts = list(self._fid_parts.values())
            base = self.fid if self.fid is not None else parts.pop(0)
            for p in parts:
                base.S += p.S
            self.fid, self._fid_parts = base, {}
        return self.fid


class PinnedU8Loader:
    """Synthetic stand-in for the decode workers of the dataset (ds_ffhq.py:307-330; PNG / zip decode is outside the path): yields
    (uint8 images [B,3,R,R] in pinned host memory, ids) for ``ids`` in order.  Item i is drawn from a generator seeded with
    (seed, i) -- the same pixels whichever rank / batch holds it; ``pool`` > 0 cycles through that many pre-drawn batches instead
    (throughput runs: drawing 12.6 MB of random bytes per batch on the host would time numpy, not the loop)."""

    def __init__(self, ids, batch_size, resolution, seed=0, pool=0):
        self.ids, self.b, self.res, self.seed, self.pool = list(ids), int(batch_size), int(resolution), seed, int(pool)
        self._cache = []

    def _draw(self, ids):
        out = torch.empty((len(ids), 3, self.res, self.res), dtype=torch.uint8)
        g = torch.Generator()
        for k, i in enumerate(ids):
            g.manual_seed(int(self.seed) * 1000003 + int(i))
            out[k].random_(0, 256, generator=g)
        return out.pin_memory() if torch.cuda.is_available() else out

    def __len__(self):
        return (len(self.ids) + self.b - 1) // self.b

    def __iter__(self):
        for n, b0 in enumerate(range(0, len(self.ids), self.b)):
            ids = self.ids[b0:b0 + self.b]
            if self.pool:
                slot = n % self.pool
                if slot >= len(self._cache):
                    self._cache.append(self._draw(ids))
                img = self._cache[slot][:len(ids)]
            else:
                img = self._draw(ids)
            yield img, ids
