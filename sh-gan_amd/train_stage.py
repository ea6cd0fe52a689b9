"""The training stage around the G / D step: phases with lazy regularisation, the per-iteration phase
loop, gradient sanitisation + all-reduce, the G_ema update.

Mirrors ``lib/experiments/stylegan_default.py``:
  * ``make_phases``  -- :304-321.  One optimiser per network; with a regularisation interval r the 'main' phase runs every
    iteration and the 'reg' phase every r-th, and the optimiser's hyper-parameters absorb the skipped steps
    (``mb_ratio = r / (r + 1)``, ``lr * mb_ratio``, ``beta ** mb_ratio``); without one a single 'both' phase.
  * ``run_phases``   -- ``train_stage.main`` :108-167: the latents of all phases drawn at once, every phase whose interval divides the
    iteration index zeroes its gradients, opens ``requires_grad`` on ITS module only, accumulates over the rounds of
    ``effective_batch_gpu`` samples (``loss.accumulate_gradients(phase, real_img, real_c, gen_z, gen_c, sync, gain=interval)``),
    sanitises the gradients (``nan_to_num(nan=0, posinf=1e5, neginf=-1e5)``) and steps.
  * ``update_ema``   -- :383-390: ``ema_beta = 0.5 ** (batch_size / max(ema_nimg, 1e-8))`` with the ``ema_rampup`` cap, buffers copied.
  * ``train``        -- the iteration loop :370-396 without its logging / snapshot / metric maintenance (``training_stats``,
    ``dnnlib`` and the snapshot pickles are not part of the reference tree; ``on_tick`` is the hook for them).
The reference wraps the networks in DistributedDataParallel (:172-187); here every phase owns a ``grad_sync.BucketedAllReduce`` over
its module's parameters (RCCL all-reduce launched from backward hooks, averaged and sanitised per bucket) -- with one rank it only
sanitises, so the same code runs on one GPU and on N."""
import os
import torch

from .grad_sync import BucketedAllReduce


class Phase:
    def __init__(self, name, module, opt, interval, sync=None):
        self.name, self.module, self.opt, self.interval, self.sync = name, module, opt, interval, sync
        self.start_event = self.end_event = None

    def __repr__(self):
        return f'Phase({self.name}, interval={self.interval})'


def make_phases(G, D, g_opt_kwargs, d_opt_kwargs, g_reg_interval=4, d_reg_interval=16, opt_class=torch.optim.Adam,
                process_group=None, bucket_bytes=64 << 20, timing=False):
    """[Gmain, Greg, Dmain, Dreg] (or Gboth / Dboth where the interval is None), stylegan_default.py:304-321."""
    phases = []
    for name, module, opt_kwargs, reg_interval in (('G', G, g_opt_kwargs, g_reg_interval), ('D', D, d_opt_kwargs, d_reg_interval)):
        params = [p for p in module.parameters()]
        kw = dict(opt_kwargs)
        sync = BucketedAllReduce(params, bucket_bytes=bucket_bytes, process_group=process_group) if any(p.is_cuda for p in params) or \
            torch.distributed.is_available() and torch.distributed.is_initialized() else None
        if reg_interval is None:
            opt = opt_class(params, **kw)
            phases.append(Phase(name + 'both', module, opt, 1, sync))
        else:                                   # lazy regularisation
            mb_ratio = reg_interval / (reg_interval + 1)
            kw['lr'] = kw['lr'] * mb_ratio
            kw['betas'] = tuple(float(beta) ** mb_ratio for beta in kw['betas'])
            opt = opt_class(params, **kw)
            phases.append(Phase(name + 'main', module, opt, 1, sync))
            phases.append(Phase(name + 'reg', module, opt, reg_interval, sync))
    if timing and torch.cuda.is_available():
        for ph in phases:
            ph.start_event, ph.end_event = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    return phases


_WARNED = set()


def _warn_unarmed(loss, phase):
    """The overlap of the gradient all-reduce with backward depends on the loss announcing the phase's LAST backward pass
    (``loss.grad_sync.arm()``, as StyleGAN2Loss._arm does for sync=True); a Loss that never does gets correct gradients, reduced
    bucket by bucket after backward.  Said once per loss class."""
    key = (type(loss).__name__, phase.name)
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(f'{type(loss).__name__} did not arm the gradient buckets in phase {phase.name}: the all-reduce runs after backward '
                      'instead of under it (call self.grad_sync.arm() before the last backward of a phase call with sync=True)')


def sanitize_(params):
    """``misc.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5, out=param.grad)`` of stylegan_default.py:160-164."""
    for p in params:
        if p.grad is not None:
            torch.nan_to_num(p.grad, nan=0.0, posinf=1e5, neginf=-1e5, out=p.grad)


def run_phases(real_img, z_dim, phases, batch_idx, loss, batch_gpu, effective_batch_gpu=None, device=None, real_c=None, gen_c=None):
    """One iteration = ``train_stage.main`` (stylegan_default.py:108-167).  real_img [batch_gpu, C, H, W]; returns the names of the
    phases that ran."""
    device = real_img.device if device is None else torch.device(device)
    eff = batch_gpu if effective_batch_gpu is None else effective_batch_gpu
    real_img = real_img.to(device).to(torch.float32)
    real_c = torch.zeros([batch_gpu, 0], device=device) if real_c is None else real_c.to(device)
    all_gen_c = torch.zeros([len(phases) * batch_gpu, 0], device=device) if gen_c is None else gen_c.to(device)
    all_gen_z = torch.randn([len(phases) * batch_gpu, z_dim]).to(device)       # drawn on the host like the reference (:128)
    all_gen_z, all_gen_c = all_gen_z.split(batch_gpu), all_gen_c.split(batch_gpu)
    ran = []
    for phase, phase_gen_z, phase_gen_c in zip(phases, all_gen_z, all_gen_c):
        if batch_idx % phase.interval != 0:
            continue
        if phase.start_event is not None:
            phase.start_event.record(torch.cuda.current_stream(device))
        _phase_body(phase, loss, real_img, real_c, phase_gen_z, phase_gen_c, eff)
        if phase.end_event is not None:
            phase.end_event.record(torch.cuda.current_stream(device))
        ran.append(phase.name)
    return ran


# Backward passes run on the calling thread.  With the engine's device worker thread, the nodes a create_graph backward creates
# (path-length / R1 regularisers) are numbered by that thread's counter and the forward nodes by the main thread's; the two advance by
# different amounts per iteration, so the ready-queue order of the double-backward graph -- and the order in which gradients with
# several consumers (x_global, the encoder's skip features) are summed -- changed from one execution to the next: identical inputs
# gave parameters that differed in the last bits (tools/ARCHIVE.md: probes/autograd_thread_order.py; every kernel is bit-repeatable).  One
# GPU per process means the worker thread bought no concurrency anyway.
SINGLE_THREADED_BACKWARD = os.environ.get('SHG_ENGINE_THREADS', '0') != '1'


def _phase_backward(phase, loss, real_img, real_c, gen_z, gen_c, eff):
    """First half of a phase (stylegan_default.py:141-158): zero the gradients, open the phase's module, run the loss's forward and
    backward passes over the rounds.  Bucket all-reduces are launched from the backward hooks (eager loop) unless ``phase.sync.defer``."""
    if phase.sync is not None:
        phase.sync.zero_grad()                       # gradients live in the all-reduce buckets
    else:
        phase.opt.zero_grad(set_to_none=True)
    phase.module.requires_grad_(True)
    if hasattr(loss, 'grad_sync'):
        loss.grad_sync = phase.sync                  # the loss arms it before the phase's LAST backward (sync=True round only)
    rr, rc, gz, gc = real_img.split(eff), real_c.split(eff), gen_z.split(eff), gen_c.split(eff)
    with torch.autograd.set_multithreading_enabled(not SINGLE_THREADED_BACKWARD):
        for round_idx in range(len(rr)):
            loss.accumulate_gradients(phase=phase.name, real_img=rr[round_idx], real_c=rc[round_idx], gen_z=gz[round_idx],
                                      gen_c=gc[round_idx], sync=(round_idx == len(rr) - 1), gain=phase.interval)
    phase.module.requires_grad_(False)
    if hasattr(loss, 'grad_sync'):
        loss.grad_sync = None


def _phase_step(phase, loss=None, reduced=False):
    """Second half (stylegan_default.py:159-166): gradients averaged over the ranks and sanitised, optimiser step.  ``reduced``: the
    buckets were all-reduced by ``phase.sync.reduce_all()`` already (split-graph form)."""
    if phase.sync is not None:
        if phase.sync.reduce and not reduced and not phase.sync.was_armed():
            _warn_unarmed(loss, phase)               # correct, but every bucket is reduced synchronously in finish(): no overlap
        phase.sync.finish(reduced=reduced)           # waits for the bucket all-reduces, averages, nan_to_num
        for p in phase.sync.untouched():             # as after zero_grad(set_to_none=True): the optimiser skips them
            p.grad = None
    else:
        sanitize_(phase.module.parameters())
    phase.opt.step()


def _phase_body(phase, loss, real_img, real_c, gen_z, gen_c, eff):
    """One phase of an iteration on given tensors: the inner block of ``run_phases`` (stylegan_default.py:141-166)."""
    _phase_backward(phase, loss, real_img, real_c, gen_z, gen_c, eff)
    _phase_step(phase, loss)


class PhaseGraphs:
    """``run_phases`` with every phase captured ONCE as a HIP graph and replayed afterwards.

    A G + D step of the FFHQ-512 networks is ~4 500 kernel launches (convolutions forward / backward / weight gradient, FIR, layer
    tails, ~2 000 small elementwise kernels of autograd glue and the optimiser); the Python + autograd + ctypes path needs ~85 ms
    to enqueue them, about what the GPU needs to run them -- the step is host-bound as soon as the kernels get faster.  A phase
    has static shapes and no host read on the device path (style-mixing cutoff, lazy-regulariser means and the Adam step counter
    live on the device: ``Adam(capturable=True)`` is required), so ``torch.cuda.graph`` can record zero_grad -> forward(s) ->
    backward(s) -> gradient sanitisation -> optimiser step as one graph per phase.  Replays read the real batch and the latents
    from static buffers that ``run`` fills first.

    Parameter-derived caches (prepared weight layouts of no-grad passes) are keyed on version counters, which a replay does not
    advance: the caches are invalidated around captures and after every replay.

    More than one rank (an active ``BucketedAllReduce``): a phase is TWO graphs with the collective between them on the host side --
    graph A = zero_grad -> forward(s) -> backward(s) into the buckets (hooks deferred: no collective is captured), then
    ``sync.reduce_all()`` (every bucket in flight at once over RCCL), then graph B = average + sanitise + optimiser step.  What is given
    up against the eager loop is the overlap of the reduction with backward (G: 5 buckets of 64 MiB, D: 4; about 2 ms of ring
    all-reduce per phase on xGMI), what is gained is the host enqueue of ~2 000 launches per phase; ``bench.py`` times both forms and
    takes the faster one on all ranks together."""

    def __init__(self, phases, loss, batch_gpu, z_dim, real_shape, device, effective_batch_gpu=None, warmup=2):
        self.phases, self.loss, self.b, self.z_dim = phases, loss, batch_gpu, z_dim
        self.device = torch.device(device)
        self.eff = batch_gpu if effective_batch_gpu is None else effective_batch_gpu
        self.real = torch.zeros(real_shape, device=self.device, dtype=torch.float32)
        self.real_c = torch.zeros([batch_gpu, 0], device=self.device)
        self.gen_c = torch.zeros([batch_gpu, 0], device=self.device)
        self.z = {ph.name: torch.zeros([batch_gpu, z_dim], device=self.device) for ph in phases}
        self.graphs = {}
        self.seen = {ph.name: 0 for ph in phases}
        self.warmup = warmup
        if warmup < 1:
            raise ValueError('PhaseGraphs: warmup >= 1 (lazily built constants, Adam state and allocator pools must exist before the capture)')
        self.split = any(ph.sync is not None and ph.sync.reduce for ph in phases)      # more than one rank: two graphs per phase
        for ph in phases:
            for g in ph.opt.param_groups:
                if not g.get('capturable', False):
                    raise ValueError('PhaseGraphs: build the optimisers with capturable=True (the step counter must live on the device)')

    def _invalidate(self):
        from .model_zoo.stylegan import _ParamCache
        _ParamCache.invalidate_all()

    def run(self, real_img, batch_idx):
        """One iteration; the first ``warmup`` runs of a phase are eager (allocator, lazily built constants), the next one is
        captured.  Returns the names of the phases that ran."""
        all_z = torch.randn([len(self.phases) * self.b, self.z_dim]).to(self.device)      # drawn on the host like the reference (:128)
        self.real.copy_(real_img.to(self.device, torch.float32))
        ran = []
        for ph, z in zip(self.phases, all_z.split(self.b)):
            if batch_idx % ph.interval != 0:
                continue
            self.z[ph.name].copy_(z)
            g = self.graphs.get(ph.name)
            if g is None and self.seen[ph.name] < self.warmup:
                self.seen[ph.name] += 1
                _phase_body(ph, self.loss, self.real, self.real_c, self.z[ph.name], self.gen_c, self.eff)
            else:
                if g is None and not self.split:
                    self._invalidate()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        _phase_body(ph, self.loss, self.real, self.real_c, self.z[ph.name], self.gen_c, self.eff)
                    self.graphs[ph.name] = g
                elif g is None:
                    # two graphs around the host-side collective.  Nothing is reduced while capturing; whether the capture worked is agreed
                    # between the ranks BEFORE the first collective of the replay (a rank that failed alone would leave the others
                    # waiting in reduce_all): all ranks raise together and the caller falls back to run_phases together.
                    self._invalidate()
                    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    err = None
                    ph.sync.defer = True
                    try:
                        with torch.cuda.graph(ga):
                            _phase_backward(ph, self.loss, self.real, self.real_c, self.z[ph.name], self.gen_c, self.eff)
                        ph.sync.defer = False
                        with torch.cuda.graph(gb, pool=ga.pool()):
                            _phase_step(ph, self.loss, reduced=True)
                    except Exception as e:              # noqa: BLE001 -- reported below, on every rank
                        err = e
                    finally:
                        ph.sync.defer = False
                    ok = torch.tensor([0.0 if err is not None else 1.0], device=self.device if torch.distributed.get_backend(ph.sync.group) == 'nccl' else 'cpu')
                    torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=ph.sync.group)
                    if float(ok.item()) < 1.0:
                        raise RuntimeError(f'PhaseGraphs: capture of phase {ph.name} failed on ' + ('this rank: ' + repr(err) if err is not None else 'another rank'))
                    g = self.graphs[ph.name] = (ga, gb)
                if self.split:
                    g[0].replay()
                    ph.sync.reduce_all()
                    g[1].replay()
                else:
                    g.replay()
                self._invalidate()
            ran.append(ph.name)
        return ran


def ema_beta(batch_size, cur_nimg, ema_kimg=10.0, ema_rampup=None):
    ema_nimg = ema_kimg * 1000
    if ema_rampup is not None:
        ema_nimg = min(ema_nimg, cur_nimg * ema_rampup)
    return 0.5 ** (batch_size / max(ema_nimg, 1e-8))


@torch.no_grad()
def update_ema(G_ema, G, batch_size, cur_nimg, ema_kimg=10.0, ema_rampup=None):
    """stylegan_default.py:383-390."""
    beta = ema_beta(batch_size, cur_nimg, ema_kimg, ema_rampup)
    for p_ema, p in zip(G_ema.parameters(), G.parameters()):
        p_ema.copy_(p.lerp(p_ema, beta))
    for b_ema, b in zip(G_ema.buffers(), G.buffers()):
        b_ema.copy_(b)
    return beta


def train(G, D, G_ema, loss, batches, phases, z_dim, batch_size, batch_gpu, total_kimg, effective_batch_gpu=None, ema_kimg=10.0,
          ema_rampup=None, kimg_per_tick=4, on_tick=None, device=None):
    """The iteration loop of stylegan_default.py:370-396: ``batches`` yields real images [batch_gpu, C, H, W] for this rank,
    ``batch_size`` is the GLOBAL batch (= batch_gpu * world).  Returns (cur_nimg, batch_idx)."""
    cur_nimg, batch_idx, cur_tick, tick_start = 0, 0, 0, 0
    for real_img in batches:
        run_phases(real_img, z_dim, phases, batch_idx, loss, batch_gpu, effective_batch_gpu, device)
        update_ema(G_ema, G, batch_size, cur_nimg, ema_kimg, ema_rampup)
        cur_nimg += batch_size
        batch_idx += 1
        done = cur_nimg >= total_kimg * 1000
        if done or cur_tick == 0 or cur_nimg >= tick_start + kimg_per_tick * 1000:
            if on_tick is not None:
                on_tick(cur_tick, cur_nimg, batch_idx)
            cur_tick += 1
            tick_start = cur_nimg
        if done:
            break
    return cur_nimg, batch_idx
