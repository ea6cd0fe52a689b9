#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): SH-GAN generator forward, images/s at 512x512 batch 16 per GPU.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic masked inputs already resident in
HBM: Generator.forward (mapping -> SHU encoder -> co-modulated synthesis, noise_mode='random' as in the
reference eval loop) + the uint8 composite.  Weights: the reference's random initialisers (seeded);
data: synthetic.  Multi-GPU = batch sharding, one process per GPU, no collective on the data path
(weak scaling); the timed region is bracketed by barrier + synchronize and the max over ranks is taken.

Rank 0 prints ONE JSON line with `roofline` (the dominant kernel class, conv_mfma: algorithmic flops /
HIP-event time measured live over the timed steps, vs the dense fp32-MFMA peak of
/opt/skills/guides/MI355X_MICROARCH.md) and `cpu_baseline` (the CPU oracle timed on this host on a
bounded sample -- a reported baseline, not the target)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PEAK_FP32_MFMA_TFLOPS = 157.3      # dense fp32 matrix peak, MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0
GFLOP_PER_IMAGE = {256: 181.6, 512: 240.9}        # SURVEY.md appendix A.3 (2*MAC), whole forward
CONV_GFLOP_PER_IMAGE = {256: 180.3, 512: 238.3}   # the 3x3 convolutions alone


def pmc_traffic(resolution, batch):
    """HBM bytes per convolution launch (conv_wino_kernel + conv_mfma_kernel, launch-weighted) from the committed PMC
    summary (tools/gpu_traffic.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command;
    FETCH_SIZE doubled per MI355X_MICROARCH.md).  PMC counters cannot be read from inside the process, so the number is
    attached from profiles/ when it matches the workload."""
    path = os.path.join(ROOT, 'profiles', f'traffic_{resolution}x{batch}.json')
    if not os.path.exists(path):
        return None
    js = json.load(open(path))
    tot = n = 0.0
    for k in ('conv_wino_kernel', 'conv_mfma_kernel'):
        d = js.get(k)
        if d:
            tot += (d['read_bytes_per_launch'] + d['write_bytes_per_launch']) * d['launches']
            n += d['launches']
    return round(tot / n / 1e9, 4) if n else None


def cpu_baseline(resolution, n_images, seed):
    """Time the CPU oracle (torch fp32 CPU ops, all host threads) on a bounded sample."""
    import torch
    from oracle import shgan_oracle as orc
    # more threads than ~32 make torch's CPU grouped convolutions *slower* on a 256-core host
    # (measured: 16 thr 0.94, 32 thr 0.97, 64 thr 0.59, 256 thr 0.04 img/s at 512x512), so cap at 32
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    sd = orc.init_state_dict(resolution, seed=seed)
    x, z, _, _ = orc.synthetic_batch(n_images, resolution, 512, seed=seed + 1)
    with torch.no_grad():
        orc.generator_forward(sd, x[:1], z[:1], resolution, noise_mode='none')      # page-in / warm-up
        t0 = time.perf_counter()
        orc.run_generator(sd, x, z, resolution, noise_mode='const')
        dt = time.perf_counter() - t0
    return dict(value=round(n_images / dt, 4), unit='images/s', cores=threads, kind='port',
                sample=f'{n_images} images {resolution}x{resolution}, 1 forward + composite after 1-image warm-up, '
                       f'oracle/shgan_oracle.py (torch CPU fp32), {dt:.1f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--resolution', type=int, default=512)
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default 16 @512, 32 @256)')
    ap.add_argument('--noise-mode', default='random')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-images', type=int, default=4)
    a = ap.parse_args()
    res = a.resolution
    batch = a.batch or {256: 32, 512: 16}.get(res, 8)

    import torch
    import torch.distributed as dist
    import shgan_amd  # noqa: F401
    from shgan_amd import eval_harness, kernels
    from test_host_logic import build_generator
    from oracle import shgan_oracle as orc      # weights only (the reference's initialisers, seeded)

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or 'RANK' in os.environ          # under torch.distributed.run even a 1-rank job joins RCCL
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    G = build_generator(res)
    G.load_state_dict(orc.init_state_dict(res, seed=0), strict=True)   # every rank initialises identically: no broadcast needed
    G = G.eval().requires_grad_(False).to(dev)
    # rank r holds its own shard of the global batch (rank-strided ids, ds_sampler.py:67)
    x, z, _, _ = eval_harness.synthetic_batch(batch, res, G.z_dim, seed=1000 + rank, device=dev, masks='bernoulli')

    def step():
        return eval_harness.run_generator(G, x, z, noise_mode=a.noise_mode)

    for _ in range(a.warmup):
        step()
    timer = kernels.KernelTimer()
    kernels.set_timer(timer)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    kernels.set_timer(None)
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out.dtype == torch.uint8 and tuple(out.shape) == (batch, 3, res, res)

    if rank == 0:
        ms = dt / a.steps * 1e3
        ips = world * batch * a.steps / dt
        tsum = timer.summary()
        zero = dict(calls=0, ms=0.0, work=0.0)
        sd, sw = tsum.get('conv_mfma', zero), tsum.get('conv_wino', zero)     # direct implicit-GEMM / Winograd F(2x2,3x3)
        conv_ms = sd['ms'] + sw['ms']
        conv_work = sd['work'] + sw['work']                 # algorithmic (direct-form) flops: 2*N*O*I*taps*pixels
        issued = sd['work'] + sw['work'] * 16.0 / 36.0      # flops the MFMA units actually execute
        ach = conv_work / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        iss = issued / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0

        def cls(d):
            return {'launches_per_step': d['calls'] // max(a.steps, 1), 'ms_per_step': round(d['ms'] / max(a.steps, 1), 3),
                    'algorithmic_tflops': round(d['work'] / (d['ms'] * 1e-3) / 1e12, 2) if d['ms'] > 0 else None}
        line = {
            'metric': 'generator images/sec', 'value': round(ips, 3), 'unit': 'images/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'FFHQ-{res} generator forward + u8 composite, random-init, batch {batch} per GPU',
                       'resolution': res, 'batch_per_gpu': batch, 'global_batch': batch * world, 'noise_mode': a.noise_mode,
                       'parallelism': f'batch-shard x{world}'},
            # achieved = ALGORITHMIC (direct-form) convolution flops / HIP-event time of every convolution launch.  The
            # stride-1 3x3 layers run as Winograd F(2x2,3x3) (16 instead of 36 multiplies per 2x2 outputs, exact fp32
            # MFMA), so `achieved` can exceed what the matrix cores execute: `mfma_issued` is the executed rate.
            'roofline': {'bound': 'mfma', 'kernel': 'conv_wino_kernel + conv_mfma_kernel (every 3x3/1x1 convolution launch)',
                         'achieved': round(ach, 3), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                         'mfma_issued': round(iss, 3), 'frac_mfma_issued': round(iss / PEAK_FP32_MFMA_TFLOPS, 4),
                         'traffic': pmc_traffic(res, batch),
                         'traffic_unit': 'GB per launch (HBM read+write, PMC FETCH_SIZE*2 + WRITE_SIZE, profiles/traffic_*.json)',
                         'launches_per_step': (sd['calls'] + sw['calls']) // max(a.steps, 1),
                         'kernel_ms_per_step': round(conv_ms / max(a.steps, 1), 3),
                         'gflop_per_step': round(conv_work / max(a.steps, 1) / 1e9, 1),
                         'classes': {'conv_wino': cls(sw), 'conv_mfma': cls(sd)},
                         'whole_forward_frac_of_fp32_mfma_peak': round(
                             ips / world * GFLOP_PER_IMAGE.get(res, 0) / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4)},
        }
        if not a.no_cpu_baseline and world == 1:
            try:
                line['cpu_baseline'] = cpu_baseline(res, a.cpu_images, seed=0)
            except Exception as e:   # the baseline is informational; never lose the GPU number over it
                line['cpu_baseline'] = {'value': None, 'unit': 'images/s', 'cores': os.cpu_count(), 'kind': 'port',
                                        'sample': f'failed: {e!r}'}
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
