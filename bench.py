#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): SH-GAN generator forward, images/s at 512x512 batch 16 per GPU.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this script spawns its own N ranks (torch.multiprocessing.spawn, one per GPU, as the
reference's main.py:83-89 does); under ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`` it
joins the ranks the launcher started (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

One "step" = one pass of the hot path over one batch of synthetic masked inputs already resident in HBM:
Generator.forward (mapping -> SHU encoder -> co-modulated synthesis, noise_mode='random' as in the reference eval
loop) + the uint8 composite.  Weights: the reference's random initialisers, seeded identically on every rank
(shgan_amd.configs.seeded_init_); data: synthetic.  Multi-GPU = batch sharding, one process per GPU, no collective on
the data path (weak scaling); the timed region is bracketed by barrier + synchronize and the max over ranks is taken.

The K timed steps are K independent batches, as in the evaluation loop of the product (eval_harness.sharded_eval): they are issued
round-robin on --pipeline-depth HIP streams (default 3; eval_harness.StreamPipeline), so that the few microseconds every one of the
~150 launch boundaries of a forward pass idles the chip are filled by the next batch's kernels.  ms_per_step = wall time / K;
the time of one batch alone on one stream is reported as pipeline.ms_per_step_single_stream (--pipeline-depth 1 makes it the headline).

The headline loop runs WITHOUT instrumentation.  A second, untimed pass then brackets every kernel launch with HIP
events on the launch stream (kernels.KernelTimer) and rank 0 prints ONE JSON line with
  roofline      the dominant kernel (conv_wino4_kernel, Winograd F(4x4,3x3)): flops the matrix cores EXECUTE (36 multiplies per
                4x4 block = 1/4 of the direct form) / HIP-event time / dense fp32-MFMA peak; the direct-form ("algorithmic")
                rate is reported beside it under its own key, per convolution class;
  hbm           the HBM-bound kernel classes: algorithmic bytes / HIP-event time / 8 TB/s;
  cpu_baseline  the CPU oracle timed on this host on a bounded sample (a reported baseline, not the target)."""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # dense fp32 matrix peak, MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense f16 matrix peak (same guide; AMD's 5 PF figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0              # HBM3E spec; 6.3 TB/s is what a float4 copy reaches (same guide)
GFLOP_PER_IMAGE = {256: 181.6, 512: 240.9}        # SURVEY.md appendix A.3 (2*MAC), whole forward, direct form


def pmc_traffic(resolution, batch, kernel='conv_wino4_kernel'):
    """HBM bytes per launch of ``kernel`` from the committed PMC summary (tools/gpu_traffic.sh: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes over this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md).  PMC counters
    cannot be read from inside the process: the number is a constant of the committed profile, attached when the profile
    matches the workload -- ``traffic_source`` in the JSON line says so."""
    path = os.path.join(ROOT, 'profiles', f'traffic_{resolution}x{batch}.json')
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path)).get(kernel)
    if not d:
        return None, None
    return round((d['read_bytes_per_launch'] + d['write_bytes_per_launch']) / 1e9, 4), os.path.relpath(path, ROOT)


def pmc_traffic_classes(resolution, batch, kernels_=('conv_wino4_kernel', 'conv_poly_up_kernel', 'conv_poly_down_kernel', 'conv_mfma_kernel',
                                                     'fir_up_march_kernel', 'fir_down_march4_kernel', 'torgb4_kernel')):
    """Per kernel class: HBM read / write GB per launch from the same committed profile (largest convolution and FIR classes)."""
    path = os.path.join(ROOT, 'profiles', f'traffic_{resolution}x{batch}.json')
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    return {k: {'launches_profiled': d[k]['launches'], 'read_GB_per_launch': round(d[k]['read_bytes_per_launch'] / 1e9, 4),
                'write_GB_per_launch': round(d[k]['write_bytes_per_launch'] / 1e9, 4)} for k in kernels_ if k in d}


def shu_floors():
    """Per-stage floor of the Spectral Hint Unit's launches from the committed probe ``profiles/r06_shu_floor.txt`` (tools/shu_floor.py: the
    product kernel beside a no-arithmetic skeleton of the same grid / workgroup / LDS footprint / global loads and stores, 400 launches back to
    back): {stage: (product_us, skeleton_us, empty_launch_us)}.  Constants of that profile (the skeleton kernels are study code, not product)."""
    path = os.path.join(ROOT, 'profiles', 'r06_shu_floor.txt')
    out = {}
    if os.path.exists(path):
        for ln in open(path):
            f = ln.split()
            if len(f) == 5 and f[0] in ('shu_rfft2', 'shu_spectral', 'shu_irfft2'):
                out[f[0]] = (float(f[1]), float(f[2]), float(f[3]))
    return out


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(resolution, bench_batch, n_images, forwards, noise_mode, seed, sweep=(16, 32, 64)):
    """the CPU restatement (oracle, torch CPU fp32) on this host, the FULL bench batch: a thread-count sweep on a
    2-image sample picks the fastest setting (more threads than ~32 make torch's CPU grouped convolutions *slower* on a 256-core
    host: 16 thr 0.94, 32 thr 0.97, 64 thr 0.59, 256 thr 0.04 img/s at 512x512, round 2), then 1 warm-up (1 image) + ``forwards`` timed
    forwards + composite of ``n_images`` images with the bench's noise mode.  The sweep result is kept in the record."""
    import torch
    from oracle import shgan_oracle as orc
    import shgan_amd  # noqa: F401
    from shgan_amd import configs
    G = configs.seeded_init_(configs.build_generator(resolution), seed=seed)
    sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    del G
    x, z, _, _ = orc.synthetic_batch(n_images, resolution, 512, seed=seed + 1)
    swept = {}
    with torch.no_grad():
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        orc.run_generator(sd, x[:1], z[:1], resolution, noise_mode=noise_mode)      # page-in / warm-up
        for th in [t for t in sweep if t <= (os.cpu_count() or 1)] or [os.cpu_count() or 1]:
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            orc.run_generator(sd, x[:2], z[:2], resolution, noise_mode=noise_mode)
            swept[th] = round(2 / (time.perf_counter() - t0), 4)
        threads = max(swept, key=swept.get)
        torch.set_num_threads(threads)
        times = []
        for _ in range(forwards):
            t0 = time.perf_counter()
            orc.run_generator(sd, x, z, resolution, noise_mode=noise_mode)
            times.append(time.perf_counter() - t0)
    best = min(times)
    return dict(value=round(n_images / best, 4), unit='images/s', cores=threads, kind='port', cpu=cpu_model(),
                host_cores=os.cpu_count(), thread_sweep_images_per_s_on_2_images=swept,
                sample=f'{forwards} timed forwards + composite (best of; all: {[round(t, 2) for t in times]} s) after a 1-image '
                       f'warm-up and a thread sweep, batch {n_images} of the bench batch {bench_batch}, {resolution}x{resolution}, noise_mode='
                       f'{noise_mode!r}, oracle/shgan_oracle.py (torch CPU fp32, {threads} threads)')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--resolution', type=int, default=512)
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default 16 @512, 32 @256)')
    ap.add_argument('--noise-mode', default='random')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-images', type=int, default=None, help='images per CPU forward (default: the full bench batch, SURVEY 8d)')
    ap.add_argument('--cpu-forwards', type=int, default=3, help='timed CPU forwards after one warm-up (SURVEY 8d: >= 3)')
    ap.add_argument('--no-second-config', action='store_true', help='skip the 256x256 batch-32 block (BASELINE config 2)')
    ap.add_argument('--no-train-step', action='store_true', help='skip the config-5 training-step block')
    ap.add_argument('--no-eval-loop', action='store_true', help='skip the config-4 evaluation-loop block')
    ap.add_argument('--eval-loop', action='store_true', help='N > 1: run the evaluation-loop block too (its all-gather / all-reduce then span the ranks)')
    ap.add_argument('--all-blocks', action='store_true',
                    help='N > 1: also run the informational second_config / train_step blocks (default: N = 1 only, so that a scaling run '
                         'measures the headline and nothing can stand between it and its JSON line)')
    ap.add_argument('--train-batch', type=int, default=8)
    ap.add_argument('--train-steps', type=int, default=3)
    ap.add_argument('--watchdog', type=int, default=0,
                    help='seconds: every rank dumps the Python stacks of all its threads to stderr every so often (faulthandler) -- says where a '
                         'multi-rank run is waiting; 0 = off')
    ap.add_argument('--graph', choices=('auto', 'on', 'off'), default='auto',
                    help='headline loop as HIP-graph replays (eval_harness.GraphPipeline) instead of eager launches; auto = on when it '
                         'captures and is not slower than the eager loop in a short trial')
    ap.add_argument('--pipeline-depth', type=int, default=None,
                    help='HIP streams the consecutive (independent) batches are issued on round-robin; 1 = one stream')
    ap.add_argument('--profile-steps', type=int, default=3, help='steps of the instrumented second pass (0 = skip)')
    ap.add_argument('--digest-out', default=None,
                    help='write {sample id: sha256 of its uint8 output} of this rank\'s last timed step to <path>.rank<r>.json and the '
                         'uint8 images to <path>.rank<r>.npy (tests/test_gpu_bench_ranks.py compares the ranks of an N = 2 run with the '
                         'single-process run id by id)')
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def max_over_ranks(dt, use_dist, backend, dev):
    import torch
    import torch.distributed as dist
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def class_tables(tsum, psteps, f16_peak=None):
    """Per kernel class of an instrumented pass (kernels.KernelTimer summary): convolution classes against the dense MFMA peak of their
    arithmetic type (executed flops: Winograd counts its transformed products), the streaming classes against 8 TB/s."""
    conv, hbm = {}, {}
    for name in sorted(tsum):
        d = tsum[name]
        if d['ms'] <= 0 or not d['calls']:
            continue
        sec = d['ms'] * 1e-3
        if name.startswith('conv'):
            peak = f16_peak if (f16_peak and 'f16' in name) else PEAK_FP32_MFMA_TFLOPS
            conv[name] = {'launches_per_step': d['calls'] // psteps, 'ms_per_step': round(d['ms'] / psteps, 3),
                          'avg_launch_us': round(d['ms'] / d['calls'] * 1e3, 1), 'executed_tflops': round(d['executed'] / sec / 1e12, 2),
                          'frac_of_mfma_peak': round(d['executed'] / sec / 1e12 / peak, 4), 'peak_tflops': peak,
                          'direct_form_tflops': round(d['work'] / sec / 1e12, 2)}
        elif name in ('upfirdn2d', 'fir_up_planar', 'torgb', 'fromrgb', 'composite_u8', 'upfirdn2d_f16', 'modtail_f16', 'relayout'):
            gbs = d['work'] / sec / 1e9
            hbm[name] = {'launches_per_step': d['calls'] // psteps, 'ms_per_step': round(d['ms'] / psteps, 3),
                         'achieved_GBps': round(gbs, 1), 'frac_of_hbm_peak': round(gbs / PEAK_HBM_GBS, 4)}
    dom = max(conv, key=lambda k: conv[k]['ms_per_step'], default=None)
    return {'mfma': conv, 'hbm': hbm, 'dominant': dom, 'dominant_frac_of_mfma_peak': conv[dom]['frac_of_mfma_peak'] if dom else None,
            'all_kernels_ms_per_step': round(sum(d['ms'] for d in tsum.values()) / psteps, 3)}


def eval_loop_block(G, res, batch, steps, a, dev, rank, world, barrier, use_dist, backend):
    """BASELINE config 4, one rank's share: the evaluation LOOP around the hot path (shgan_default.py:264-300) --
    per batch: decoded uint8 images from the loader (pinned host memory; a pool of pre-drawn batches stands in for the decode workers)
    -> H2D on the copy stream -> freeform masks drawn on the device from the reference's numpy draws -> input assembly -> z ~ N(0,1) ->
    G + uint8 composite into the result buffer -> stand-in [B,2048] features (the Inception detector is a download) -> fp64 FID
    moments; at the end one all-reduce of the moments and one all-gather + re-interleave of the uint8 results.  Timed barrier to
    barrier with everything above inside; max over ranks."""
    import numpy as np
    import torch
    from shgan_amd import eval_harness
    n_items = world * batch * steps

    def make(n):
        return eval_harness.EvalLoop(G, dev, res, n, rank=rank, world=world, noise_mode=a.noise_mode, seed=0, depth=a.pipeline_depth,
                                     feature_fn=eval_harness.standin_features, timing=True)
    warm = make(world * batch * 4)
    pool = eval_harness.PinnedU8Loader(warm.ids, batch, res, seed=1000, pool=4)
    np.random.seed(1000 + rank)
    warm.run(pool)
    warm.gather()
    torch.cuda.synchronize()
    del warm
    loop = make(n_items)
    loader = eval_harness.PinnedU8Loader(loop.ids, batch, res, seed=1000, pool=4)
    loader._cache = pool._cache                            # the pinned pool drawn during the warm-up
    np.random.seed(2000 + rank)                            # mask draws: numpy's global generator, as the reference's workers
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.run(loader)
    t_issue = time.perf_counter() - t0                     # host side of the loop: everything enqueued (the device may still be busy)
    torch.cuda.synchronize()
    t_loop = time.perf_counter() - t0
    images, fid = loop.gather()
    torch.cuda.synchronize()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, use_dist, backend, dev)
    n_fid = float(fid.S[fid.dim, fid.dim].item())
    # steady state: completion time of batch k on the device (timing events on the batches' streams), first / last quarter left out
    evs = loop.batch_done_events
    steady = None
    if len(evs) >= 12:
        k0, k1 = len(evs) // 4, len(evs) - 1 - len(evs) // 4
        steady = evs[k0].elapsed_time(evs[k1]) / (k1 - k0)
    ok = tuple(images.shape) == (n_items, 3, res, res) and images.dtype == torch.uint8
    del loop, images, fid
    torch.cuda.empty_cache()
    return {'workload': f'Places2/FFHQ-{res} evaluation loop, one rank\'s share: {steps} batches of {batch} per GPU -- uint8 loader -> H2D (copy stream) -> '
                        'device freeform masks -> assemble -> G + u8 composite (streamed into the result buffer) -> stand-in features '
                        '[B,2048] -> fp64 FID moments; then all-reduce(moments) + all-gather(u8) + zipzap',
            'images_per_s': round(n_items / dt, 3), 'ms_per_batch': round(dt / steps * 1e3, 3), 'batches': steps, 'n_gpus': world,
            'ms_per_batch_loop_only': round(t_loop / steps * 1e3, 3), 'host_issue_ms_per_batch': round(t_issue / steps * 1e3, 3),
            'gather_and_reduce_ms': round((dt - t_loop) * 1e3, 3), 'fid_samples_counted': n_fid, 'result_ok': bool(ok),
            'steady_state_ms_per_batch': round(steady, 3) if steady else None,
            'steady_state_images_per_s': round(world * batch / steady * 1e3, 1) if steady else None,
            'steady_state_note': 'device completion times of the middle half of the batches (rank 0): a real evaluation runs hundreds of batches per '
                                 'rank, the whole-loop figure above carries the start (host draws the first masks before the device has work) and the drain',
            'not_in_the_loop': 'PNG / zip decode (dataset workers) and the Inception-v3 detector (a download): a pinned pool of pre-drawn '
                               'uint8 batches and a fixed linear map stand in for them'}


def forward_block(res, batch, steps, warmup, a, dev, rank, world, barrier, use_dist, backend, fp16=False):
    """The headline loop on another configuration (BASELINE config 2: FFHQ-256 batch 32): same pipeline, same barrier / max-over-ranks timing."""
    import torch
    from shgan_amd import configs, eval_harness
    kw = dict(use_fp16_before_res=64, use_fp16_after_res=32) if fp16 else {}
    G = configs.seeded_init_(configs.build_generator(res, **kw), seed=0).eval().requires_grad_(False).to(dev)
    ids = [rank + world * k for k in range(batch)]
    x, z, _, _ = eval_harness.synthetic_items(ids, res, G.z_dim, seed=1000, device=dev)

    def step():
        return eval_harness.run_generator(G, x, z, noise_mode=a.noise_mode)
    pipe = eval_harness.StreamPipeline(dev, depth=a.pipeline_depth)
    for _ in range(warmup):
        pipe.run(step)
    pipe.join()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.run(step)
    pipe.join()
    torch.cuda.synchronize()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, use_dist, backend, dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    lat = (time.perf_counter() - t1) / 3 * 1e3
    classes = None
    if rank == 0 and a.profile_steps > 0:                 # the same instrumented pass as the headline's: HIP events around every launch
        from shgan_amd import kernels
        timer = kernels.KernelTimer()
        kernels.set_timer(timer)
        for _ in range(a.profile_steps):
            step()
        torch.cuda.synchronize()
        kernels.set_timer(None)
        classes = class_tables(timer.summary(), a.profile_steps, PEAK_F16_MFMA_TFLOPS if fp16 else None)
    barrier()
    del G, x, z
    torch.cuda.empty_cache()
    return {'workload': f'FFHQ-{res} generator forward + u8 composite, random-init, batch {batch} per GPU'
                        + (', the reference\'s use_fp16 blocks on (encoder > 64, synthesis > 32: NHWC fp16-MFMA kernels, fused layer tails); NOT the '
                           'shipped configuration (use_fp16_*: null) and not the headline' if fp16 else ''),
            'dtype': 'f16 blocks + f32' if fp16 else 'f32', 'value': round(world * batch * steps / dt, 3),
            'unit': 'images/s', 'ms_per_step': round(dt / steps * 1e3, 3), 'steps': steps, 'warmup': warmup, 'n_gpus': world,
            'ms_per_step_single_stream': round(lat, 3), 'gflop_per_image_direct_form': GFLOP_PER_IMAGE.get(res),
            'roofline_classes': classes}


def train_block(a, dev, rank, world, barrier, use_dist, backend, fp16=False):
    """BASELINE config 5: FFHQ-512 G + D training step (stylegan_default_loss.py:53-128 on the co-modulated generator, losses.InpaintingLoss),
    batch ``--train-batch`` per GPU, fp32, Adam, noise_mode='random', style mixing 0.9, gradients in RCCL all-reduce buckets (world > 1).
    A step = Gmain + Dmain (every iteration); the lazy regularisers Greg (every 4th) / Dreg (every 16th) are timed in one full iteration and
    reported amortised.  Not part of the headline; one extra instrumented iteration gives the per-class kernel times."""
    import torch
    from shgan_amd import configs, kernels, losses, train_stage as ts
    from shgan_amd.model_zoo import stylegan
    b, res = a.train_batch, 512
    torch.manual_seed(1234)                              # identical initial weights on every rank
    # fp16: the reference's `use_fp16` blocks on the four highest resolutions (NHWC fp16-MFMA kernels, fp32 accumulation, fp32 master weights);
    # the encoder keeps its 64^2 block float32 because that feature feeds the float32 SHU
    G = configs.seeded_init_(configs.build_generator(res, **(dict(use_fp16_before_res=64, use_fp16_after_res=32) if fp16 else {})),
                             seed=0).to(dev).train().requires_grad_(False)
    D = stylegan.Discriminator(resolution=res, ic_n=4, ch_base=32768, ch_max=512, use_fp16_before_res=(32 if fp16 else None), mbstd_group_size=4,
                               mbstd_c_n=1).to(dev).train().requires_grad_(False)
    torch.manual_seed(4321 + rank)                       # per-rank data / latents / noise
    real = torch.rand(b, 3, res, res, device=dev) * 2 - 1
    mask = (torch.rand(b, 1, res, res, device=dev) < 0.7).float()
    real4 = torch.cat([mask - 0.5, real], dim=1)
    L = losses.InpaintingLoss(dev, G, D, composite_fake=True, noise_mode='random', style_mixing_prob=0.9, r1_gamma=10, pl_batch_shrink=2, pl_weight=2)
    # phases as HIP graphs (train_stage.PhaseGraphs); eager loop timed beside it.  N > 1: two graphs per phase around the host-side bucket
    # all-reduce (the eager loop overlaps the reduction with backward instead); the faster form is taken, on all ranks together
    use_graph = a.graph != 'off'
    kw = dict(lr=0.002, betas=(0.0, 0.99), eps=1e-8, capturable=use_graph, fused=True)     # one multi-tensor kernel per optimiser step
    phases = ts.make_phases(G, D, kw, kw, g_reg_interval=4, d_reg_interval=16)
    pg = ts.PhaseGraphs(phases, L, b, 512, tuple(real4.shape), dev) if use_graph else None
    mode = {'graph': False}

    def iteration(idx):
        if mode['graph']:
            return pg.run(real4, idx)
        return ts.run_phases(real4, 512, phases, batch_idx=idx, loss=L, batch_gpu=b, device=dev)

    def timed(idxs):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in idxs:
            iteration(i)
        torch.cuda.synchronize()
        barrier()
        return max_over_ranks(time.perf_counter() - t0, use_dist, backend, dev) / len(idxs) * 1e3
    iteration(0)                                          # warm-up: all four phases (weight-layout caches, allocator)
    iteration(1)
    ms_main = timed([1 + (k % 3) for k in range(a.train_steps)])       # Gmain + Dmain only (batch_idx % 4 != 0)
    ms_greg = timed([4, 4])                               # Gmain + Greg + Dmain (batch_idx % 4 == 0, % 16 != 0)
    ms_all = timed([0])                                   # Gmain + Greg + Dmain + Dreg
    graph_info = {'used': False}
    if use_graph:
        eager = (ms_main, ms_greg, ms_all)
        try:
            mode['graph'] = True
            for _ in range(3):
                iteration(0)                              # two eager runs of every phase inside PhaseGraphs, then the capture + first replay
            torch.cuda.synchronize()
            g_main = timed([1 + (k % 3) for k in range(a.train_steps)])
            g_greg = timed([4, 4])
            g_all = timed([0])
            # (the times are max-over-ranks already: every rank sees the same numbers and takes the same branch)
            faster = world == 1 or g_main <= eager[0]
            graph_info = {'used': bool(faster), 'eager_ms_per_step': round(eager[0], 2), 'eager_ms_iteration_with_both_lazy_regularisers': round(eager[2], 2),
                          'graph_ms_per_step': round(g_main, 2), 'split_around_all_reduce': bool(pg.split),
                          'note': 'every phase captured once as a HIP graph (train_stage.PhaseGraphs) and replayed; the eager loop '
                                  '(Python + autograd + ctypes enqueue of ~4 500 launches per step) timed beside it'
                                  + ('; N > 1: two graphs per phase with the bucket all-reduce between them (no overlap with backward), the faster of '
                                     'the two forms reported' if pg.split else '')}
            if faster:
                ms_main, ms_greg, ms_all = g_main, g_greg, g_all
        except Exception as e:
            graph_info = {'used': False, 'error': repr(e)[:400]}
            ms_main, ms_greg, ms_all = eager
        mode['graph'] = False
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
    finite = all(bool(torch.isfinite(v).all()) for v in L.stats.values())
    cls = {}
    # the instrumented iteration runs on EVERY rank (its gradient all-reduce is a collective: until round 6 only rank 0 ran it and waited
    # for the others, who stood in the barrier below -- found on the test box with `--gpus 2 --all-blocks --watchdog 120`); only rank 0 times
    timer = kernels.KernelTimer() if rank == 0 else None
    kernels.set_timer(timer)
    iteration(1)
    torch.cuda.synchronize()
    kernels.set_timer(None)
    if rank == 0:
        for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]['ms']):
            if v['ms'] <= 0:
                continue
            e = {'ms_per_step': round(v['ms'], 3), 'launches': v['calls']}
            if k.startswith('conv'):
                sec = v['ms'] * 1e-3
                e.update(direct_form_tflops=round(v['work'] / sec / 1e12, 2), executed_tflops=round(v['executed'] / sec / 1e12, 2))
                if 'f16' in k:
                    e['frac_of_f16_mfma_peak'] = round(v['executed'] / sec / 1e12 / PEAK_F16_MFMA_TFLOPS, 4)
                else:
                    e['frac_of_fp32_mfma_peak'] = round(v['executed'] / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
            cls[k] = e
    barrier()
    mem = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    for ph in phases:
        if ph.sync is not None:
            ph.sync.remove()
    del G, D, L, phases, pg
    torch.cuda.empty_cache()
    # Greg runs every 4th and Dreg every 16th iteration: (ms_all - ms_main) is their joint cost in an iteration where both run
    # the reference's iteration (stylegan_default.py:304-321): Gmain + Dmain every time, Greg every 4th, Dreg every 16th -- the amortised
    # iteration is the training figure; the phases it is made of are listed beside it
    t_greg, t_dreg = max(ms_greg - ms_main, 0.0), max(ms_all - ms_greg, 0.0)
    ms_amort = ms_main + t_greg / 4 + t_dreg / 16
    return {'workload': f'FFHQ-512 G + D training iteration (Gmain + Dmain: non-saturating logistic loss, Adam; lazy regularisers amortised), '
                        f'random-init, batch {b} per GPU, '
                        + ('fp16 blocks at the four highest resolutions (fp16 MFMA, fp32 accumulation / master weights)' if fp16 else 'fp32'),
            'ms_per_iteration_amortised': round(ms_amort, 2), 'images_per_s': round(world * b / ms_amort * 1e3, 2),
            'phases_ms': {'Gmain_plus_Dmain': round(ms_main, 2), 'Greg_path_length_batch_half': round(t_greg, 2), 'Dreg_R1': round(t_dreg, 2),
                          'amortisation': 'Gmain + Dmain + Greg / 4 + Dreg / 16'},
            'ms_per_step': round(ms_main, 2), 'images_per_s_main_phases_only': round(world * b / ms_main * 1e3, 2), 'steps': a.train_steps, 'n_gpus': world,
            'ms_iteration_with_both_lazy_regularisers': round(ms_all, 2),
            'lazy_regularisers': 'Greg (path length, batch/2) every 4th, Dreg (R1) every 16th iteration (stylegan_default.py:304-321)',
            'objective': 'stylegan_default_loss.py:53-128 with the generator output composited with the known pixels before the critic sees it '
                         '(InpaintingLoss composite_fake=True, the objective to train an inpainting generator with; rounds 3-5 timed the raw-output '
                         'form: one more elementwise pass per generator call); Dmain judges fake + real as one stacked critic pass (same logits and gradients)',
            'dtype': 'f16 blocks + f32' if fp16 else 'f32', 'losses_finite': finite, 'peak_memory_GiB': round(mem, 1),
            'grad_all_reduce': (backend if world > 1 else None), 'hip_graph': graph_info, 'kernel_classes_one_step_rank0': cls,
            'conv_kernel_ms': round(sum(v['ms_per_step'] for k, v in cls.items() if k.startswith('conv')), 2) if cls else None}


def worker(local_rank, a, spawned_world=None, port=None):
    """One rank.  ``spawned_world`` is set when this process was started by bench.py's own spawn."""
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL / device-memory sharing across processes needs it on this driver
    if a.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(a.watchdog, repeat=True, file=sys.stderr)
    import torch
    import torch.distributed as dist
    import shgan_amd  # noqa: F401
    from shgan_amd import configs, eval_harness, kernels

    if spawned_world is not None:
        os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(spawned_world),
                          MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    res = a.resolution
    batch = a.batch or {256: 32, 512: 16}.get(res, 8)
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit('bench.py needs a HIP device: the product has no CPU path')
    oversub = ndev < world               # fewer GPUs than ranks (launch-path smoke on a 1-GPU box): ranks share devices
    torch.cuda.set_device(local % ndev)
    dev = torch.device('cuda', local % ndev)
    use_dist = world > 1 or 'RANK' in os.environ          # under torch.distributed.run even a 1-rank job joins RCCL
    backend = None
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        backend = 'gloo' if oversub else 'nccl'            # RCCL needs one device per rank
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)

    def barrier():
        if use_dist:
            if backend == 'nccl':
                dist.barrier(device_ids=[dev.index])
            else:
                dist.barrier()

    G = configs.build_generator(res)
    configs.seeded_init_(G, seed=0)        # every rank initialises identically: no weight broadcast needed
    G = G.eval().requires_grad_(False).to(dev)
    # rank r holds its own shard of the global batch (rank-strided ids, ds_sampler.py:67)
    ids = [rank + world * k for k in range(batch)]
    x, z, _, _ = eval_harness.synthetic_items(ids, res, G.z_dim, seed=1000, device=dev)
    torch.manual_seed(world + rank)        # per-rank noise stream (shgan_default.py:165-167)

    def step():
        return eval_harness.run_generator(G, x, z, noise_mode=a.noise_mode)

    if a.pipeline_depth is None:
        a.pipeline_depth = eval_harness.PIPELINE_DEPTH

    def agree(flag):
        """every rank takes the same loop: the flag holds only if it holds on all ranks"""
        if use_dist:
            t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            flag = bool(t.item() > 0.5)
        return flag

    def loop(pipe, n, graphed):
        o = None
        for _ in range(n):
            o = pipe.run(x, z) if graphed else pipe.run(step)
        pipe.join()
        return o

    # ---- eager loop (round-robin streams) or the same loop as HIP-graph replays
    pipe = eval_harness.StreamPipeline(dev, depth=a.pipeline_depth)
    loop(pipe, max(a.warmup, 1), False)
    graph_info = {'mode': a.graph, 'used': False}
    gpipe = None
    if a.graph != 'off':
        try:
            gpipe = eval_harness.GraphPipeline(dev, lambda x_, z_: eval_harness.run_generator(G, x_, z_, noise_mode=a.noise_mode), (x, z),
                                               depth=a.pipeline_depth, watch=list(G.parameters()) + list(G.buffers()))
            loop(gpipe, max(a.pipeline_depth, 2), True)
            torch.cuda.synchronize()
        except Exception as e:
            graph_info['capture_error'] = repr(e)[:300]
            gpipe = None
        captured = agree(gpipe is not None)
        graph_info['captured_on_all_ranks'] = captured
        use_graph = captured
        if captured and a.graph == 'auto':
            trial = {}
            for name, pp, gr in (('eager', pipe, False), ('graph', gpipe, True), ('eager', pipe, False), ('graph', gpipe, True)):
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                loop(pp, 6, gr)
                torch.cuda.synchronize()
                trial[name] = min(trial.get(name, 1e9), (time.perf_counter() - t_) / 6 * 1e3)
            graph_info['trial_ms_per_step'] = {k: round(v, 3) for k, v in trial.items()}
            # one GPU: the replay must not be slower; several ranks on one host: prefer it unless clearly slower (eight interpreters
            # enqueueing ~120 launches per step compete for the host's cores -- the replay is one call)
            use_graph = agree(trial['graph'] <= trial['eager'] * (1.0 if world == 1 else 1.03))
        graph_info['used'] = use_graph
        if not use_graph:
            gpipe = None
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = loop(gpipe, a.steps, True) if gpipe is not None else loop(pipe, a.steps, False)
    t_issue = time.perf_counter() - t0                  # host side: K steps enqueued (nothing in the loop waits for the device)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, use_dist, backend, dev)
    rank_ms = [dt_local / a.steps * 1e3]
    if use_dist:                           # every rank's own loop time: a straggler shows in the line the first time N > 1 runs on hardware
        t = torch.zeros(world, dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        t[rank] = dt_local / a.steps * 1e3
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rank_ms = [float(v) for v in t.tolist()]
    issue_ms = [t_issue / a.steps * 1e3]
    if use_dist:                           # every rank's host enqueue time per step: N interpreters on one host compete for cores
        t = torch.zeros(world, dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        t[rank] = t_issue / a.steps * 1e3
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        issue_ms = [float(v) for v in t.tolist()]
    ranks_counted = 1
    if use_dist:                           # every rank contributes 1: the line proves how many ranks the collective really spanned
        t = torch.ones(1, dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ranks_counted = int(t.item())
    assert out.dtype == torch.uint8 and tuple(out.shape) == (batch, 3, res, res)
    if a.digest_out:
        import hashlib
        import numpy as np
        o8 = out.cpu().numpy()
        with open(f'{a.digest_out}.rank{rank}.json', 'w') as fh:
            json.dump({str(i): hashlib.sha256(o8[k].tobytes()).hexdigest() for k, i in enumerate(ids)}, fh)
        # the images themselves: a sample's output depends on its batch-mates at the 1e-6 level through the batch-global style RMS of
        # stylegan.py:147, so digests of one id agree across batch compositions only up to a few truncation flips
        np.save(f'{a.digest_out}.rank{rank}.npy', o8)
    # one batch alone on one stream (latency of a step; not the headline)
    lat_ms = None
    if rank == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        lat_ms = (time.perf_counter() - t1) / 3 * 1e3
    barrier()

    # ---- second pass (not part of the headline): per-kernel-class HIP-event times
    tsum = {}
    psteps = max(0, a.profile_steps)
    if rank == 0 and psteps:
        timer = kernels.KernelTimer()
        kernels.set_timer(timer)
        for _ in range(psteps):
            step()
        torch.cuda.synchronize()
        kernels.set_timer(None)
        tsum = timer.summary()
    barrier()
    del pipe, gpipe
    torch.cuda.empty_cache()
    second = train = fp16_eval = evloop = None
    extra = world == 1 or a.all_blocks
    if (extra or a.eval_loop) and not a.no_eval_loop:
        try:
            evloop = eval_loop_block(G, res, batch, a.steps, a, dev, rank, world, barrier, use_dist, backend)
        except Exception as e:             # informational block: never lose the headline over it
            if world > 1:
                raise
            evloop = {'error': repr(e)[:400]}
    if extra and not a.no_second_config and res == 512 and a.batch is None:
        second = forward_block(256, 32, max(10, a.steps // 2), 6, a, dev, rank, world, barrier, use_dist, backend)
        fp16_eval = forward_block(512, 16, max(10, a.steps // 2), 6, a, dev, rank, world, barrier, use_dist, backend, fp16=True)
    if extra and not a.no_train_step and res == 512 and a.batch is None:
        del G, x, z, out
        torch.cuda.empty_cache()
        try:
            train = train_block(a, dev, rank, world, barrier, use_dist, backend)
            train['fp16_blocks'] = train_block(a, dev, rank, world, barrier, use_dist, backend, fp16=True)
        except Exception as e:             # informational block: never lose the headline over it
            if world > 1:
                raise                      # (a rank that dropped out of a collective would hang the others)
            train = {'error': repr(e)}

    if rank == 0:
        ms = dt / a.steps * 1e3
        ips = world * batch * a.steps / dt

        def cls_conv(name):
            d = tsum.get(name)
            if not d or d['ms'] <= 0:
                return None
            sec = d['ms'] * 1e-3
            return {'launches_per_step': d['calls'] // psteps, 'ms_per_step': round(d['ms'] / psteps, 3),
                    'avg_launch_us': round(d['ms'] / d['calls'] * 1e3, 1),
                    'executed_tflops': round(d['executed'] / sec / 1e12, 2),
                    'frac_of_fp32_mfma_peak': round(d['executed'] / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    'direct_form_tflops': round(d['work'] / sec / 1e12, 2)}

        def cls_hbm(name):
            d = tsum.get(name)
            if not d or d['ms'] <= 0:
                return None
            gbs = d['work'] / (d['ms'] * 1e-3) / 1e9
            return {'launches_per_step': d['calls'] // psteps, 'ms_per_step': round(d['ms'] / psteps, 3),
                    'algorithmic_GB_per_step': round(d['work'] / psteps / 1e9, 3), 'achieved_GBps': round(gbs, 1),
                    'frac_of_hbm_peak': round(gbs / PEAK_HBM_GBS, 4)}

        conv = {k: v for k, v in ((n_, cls_conv(n_)) for n_ in sorted(tsum) if n_.startswith('conv_')) if v}
        hbm = {k: v for k, v in ((n, cls_hbm(n)) for n in ('upfirdn2d', 'fir_up_planar', 'torgb', 'fromrgb', 'composite_u8'))
               if v}

        def shu_block():
            # the Spectral Hint Unit's three launches, each against the roof that bounds it: the spectral stage (1x1 conv + ReLU + band
            # filter: 64-wide GEMMs per spectral pixel) is matrix work on the fp32 MFMA; the two transform stages are short dependent
            # chains per 64x64 plane (row DFT, column DFT as MFMA passes with a barrier between) on a few MB -- latency-bound: their time
            # is the figure, the byte rate is listed only to show they are nowhere near HBM
            st = {}
            floors = shu_floors()
            for name, kind in (('shu_rfft2', 'latency'), ('shu_spectral', 'mfma'), ('shu_irfft2', 'latency')):
                d = tsum.get(name)
                if not d or d['ms'] <= 0:
                    continue
                e = {'us_per_step': round(d['ms'] / psteps * 1e3, 1), 'launches_per_step': d['calls'] // psteps, 'bound': kind}
                if kind == 'mfma':
                    tf = d['work'] / (d['ms'] * 1e-3) / 1e12
                    e.update(executed_tflops=round(tf, 2), frac_of_fp32_mfma_peak=round(tf / PEAK_FP32_MFMA_TFLOPS, 4))
                else:
                    e['algorithmic_GBps'] = round(d['work'] / (d['ms'] * 1e-3) / 1e9, 1)
                if name in floors:
                    pr, sk, em = floors[name]
                    e.update(floor_us=sk, empty_launch_us=em, back_to_back_us=pr, frac_of_floor=round(sk / pr, 3))
                st[name] = e
            if not st:
                return None
            return {'ms_per_step': round(sum(v['us_per_step'] for v in st.values()) / 1e3, 4), 'stages': st,
                    'floor_source': 'profiles/r06_shu_floor.txt' if floors else None,
                    'floor_note': 'floor_us = a skeleton launch of the same grid / LDS / load-store footprint without arithmetic, back_to_back_us = the product '
                                  'kernel in the same probe (no per-launch events), frac_of_floor = floor / product there; us_per_step is this run\'s '
                                  'HIP-event time inside the generator (launch gaps included)' if floors else None}
        shu = shu_block()
        conv_ms = sum(tsum[k]['ms'] for k in tsum if k.startswith('conv_'))
        conv_exec = sum(tsum[k]['executed'] for k in tsum if k.startswith('conv_'))
        conv_alg = sum(tsum[k]['work'] for k in tsum if k.startswith('conv_'))
        # the dominant kernel = the convolution class with the largest share of the step
        DOM = {'conv_wino4': ('conv_wino4_kernel', 'Winograd F(4x4,3x3), 3x3 stride-1 layers: 36/144 of the direct-form flops'),
               'conv_wino': ('conv_wino_kernel', 'Winograd F(2x2,3x3), 3x3 stride-1 layers: 16/36 of the direct-form flops')}
        dom_name = max((k for k in conv if k in DOM), key=lambda k: conv[k]['ms_per_step'], default=None)
        dom = conv.get(dom_name)
        traffic, traffic_src = pmc_traffic(res, batch, DOM[dom_name][0]) if dom_name else (None, None)
        roof = {'bound': 'mfma', 'kernel': f'{DOM[dom_name][0]} ({DOM[dom_name][1]}; the largest share of a step)' if dom_name else None,
                'achieved': dom['executed_tflops'] if dom else None, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': dom['frac_of_fp32_mfma_peak'] if dom else None,
                'definition': 'flops executed on the matrix cores (Winograd: the multiplies of the transformed products, not the direct-form '
                              'count) / HIP-event time of the launches in the instrumented pass / dense fp32 MFMA peak (157.3 TFLOP/s at '
                              '2.4 GHz; under this load the chip sustains about 2.1 GHz)',
                'avg_launch_us': dom['avg_launch_us'] if dom else None,
                'direct_form_tflops': dom['direct_form_tflops'] if dom else None,
                'traffic': traffic, 'traffic_unit': 'GB per launch (HBM read+write, PMC FETCH_SIZE*2 + WRITE_SIZE)',
                'traffic_source': f'constant of the committed profile {traffic_src}, not measured in this run' if traffic_src else None,
                'traffic_classes': pmc_traffic_classes(res, batch),
                'classes': conv,
                'all_conv': {'ms_per_step': round(conv_ms / psteps, 3) if psteps else None,
                             'executed_tflops': round(conv_exec / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None,
                             'frac_of_fp32_mfma_peak': round(conv_exec / (conv_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if conv_ms else None,
                             'direct_form_tflops': round(conv_alg / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None},
                'whole_step': {'executed_gflop_per_step': round(conv_exec / psteps / 1e9, 1) if psteps else None,
                               'executed_tflops_over_wall': round(conv_exec / psteps / (ms * 1e-3) / 1e12, 2) if psteps else None,
                               'frac_of_fp32_mfma_peak': round(conv_exec / psteps / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if psteps else None,
                               'direct_form_gflop_per_image': GFLOP_PER_IMAGE.get(res)}}
        line = {
            'metric': 'generator images/sec', 'value': round(ips, 3), 'unit': 'images/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'FFHQ-{res} generator forward + u8 composite, random-init, batch {batch} per GPU',
                       'resolution': res, 'batch_per_gpu': batch, 'global_batch': batch * world, 'noise_mode': a.noise_mode,
                       'parallelism': f'batch-shard x{world}', 'launcher': 'self-spawn' if spawned_world else
                       ('torch.distributed.run' if 'RANK' in os.environ and spawned_world is None and use_dist else 'single process'),
                       'collective_backend': backend, 'ranks_all_reduced': ranks_counted, 'ranks_share_devices': oversub,
                       'stream_pipeline_depth': a.pipeline_depth, 'hip_graph': graph_info,
                       'ms_per_step_by_rank': {'min': round(min(rank_ms), 3), 'max': round(max(rank_ms), 3),
                                               'all': [round(v, 3) for v in rank_ms],
                                               'note': 'each rank\'s own loop time (before the closing barrier); ms_per_step = barrier-to-barrier, max over ranks'},
                       'host_enqueue_ms_per_step_by_rank': {'min': round(min(issue_ms), 3), 'max': round(max(issue_ms), 3),
                                                            'all': [round(v, 3) for v in issue_ms],
                                                            'note': 'wall time a rank\'s interpreter needs to enqueue one step (eager: ~150 ctypes '
                                                                    'launches; graph: one replay); the step is host-bound where this reaches ms_per_step'}},
            'pipeline': {'depth': a.pipeline_depth, 'ms_per_step_single_stream': round(lat_ms, 3) if lat_ms else None,
                         'note': 'the K timed steps are independent batches issued round-robin on `depth` HIP streams '
                                 '(eval_harness.StreamPipeline, the evaluation loop of the product): the launch-boundary gaps of '
                                 'one batch are filled by the kernels of the next; ms_per_step = wall time / K (throughput), '
                                 'ms_per_step_single_stream = one batch alone (latency)'},
            'timing': 'headline loop uninstrumented; roofline/hbm from a separate instrumented pass of '
                      f'{psteps} steps (HIP events on the launch stream)',
            'roofline': roof,
            'hbm': {'bound': 'hbm', 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'classes': hbm},
            'shu': shu,
            'eval_loop': evloop,
            'second_config': second,
            'fp16_blocks_eval': fp16_eval,
            'train_step': train,
        }
        if not a.no_cpu_baseline and world == 1:
            try:
                line['cpu_baseline'] = cpu_baseline(res, batch, a.cpu_images or batch, a.cpu_forwards, a.noise_mode, seed=0)
                if second is not None:
                    line['cpu_baseline']['second_config'] = cpu_baseline(256, 32, a.cpu_images or 32, a.cpu_forwards, a.noise_mode, seed=0)
            except Exception as e:   # the baseline is informational; never lose the GPU number over it
                line['cpu_baseline'] = {'value': None, 'unit': 'images/s', 'cores': os.cpu_count(), 'kind': 'port',
                                        'sample': f'failed: {e!r}'}
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def main():
    a = parse()
    if 'RANK' in os.environ:                         # started by torch.distributed.run (or any external launcher)
        world = int(os.environ.get('WORLD_SIZE', 1))
        if world != a.gpus:
            raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
        worker(int(os.environ.get('LOCAL_RANK', 0)), a)
    elif a.gpus == 1:
        worker(0, a)
    else:                                            # own launcher: one process per GPU
        import torch.multiprocessing as mp
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        mp.spawn(worker, args=(a, a.gpus, free_port()), nprocs=a.gpus, join=True)


if __name__ == '__main__':
    main()
