"""The multi-rank path of ``bench.py`` on ONE device (the driver's GPU test box has a single MI355X): ``--gpus 2`` spawns two ranks that
share the GPU (``ranks_share_devices``; the process group falls back to gloo because RCCL needs a device per rank), so the spawn, the
all-reduced rank count, the per-rank step times, the all-reduced HIP-graph decision and the rank-strided sample ids
(lib/data_factory/common/ds_sampler.py:58-68, main.py:86-89, lib/utils.py:304-309) all execute here.  The two ``nccl`` tests of
tests/test_gpu_multi.py keep activating themselves wherever two GPUs exist."""
import json
import os

import numpy as np
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

COMMON = ['--steps', '2', '--warmup', '1', '--no-train-step', '--no-second-config', '--no-cpu-baseline', '--no-eval-loop', '--profile-steps', '0',
          '--resolution', '256', '--batch', '4', '--noise-mode', 'const']


def _bench(tmp_path, tag, gpus, common=None):
    dig = str(tmp_path / tag)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--digest-out', dig] + (common or COMMON), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    line = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')][-1])
    digests, images = {}, {}
    for r in range(gpus):
        with open(f'{dig}.rank{r}.json') as fh:
            digests[r] = json.load(fh)
        images[r] = dict(zip(digests[r].keys(), np.load(f'{dig}.rank{r}.npy')))
    return line, digests, images


def _same_image(a, b):
    """One sample id processed beside different batch-mates: the batch-global style RMS of stylegan.py:147 couples the samples of a batch at
    the 1e-6 level (cancels under demodulation up to rounding), so a few uint8 truncations may flip by one step."""
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return int(d.max()) <= 1 and float((d > 0).mean()) < 1e-3


def test_bench_two_ranks_share_one_device(tmp_path):
    line2, dig2, img2 = _bench(tmp_path, 'two', 2)
    cfg = line2['config']
    assert line2['n_gpus'] == 2 and cfg['global_batch'] == 8 and cfg['launcher'] == 'self-spawn'
    assert cfg['ranks_all_reduced'] == 2 and cfg['ranks_share_devices'] and cfg['collective_backend'] == 'gloo'
    assert len(cfg['ms_per_step_by_rank']['all']) == 2 and all(v > 0 for v in cfg['ms_per_step_by_rank']['all'])
    # the graph-versus-eager decision was taken, and taken together (all-reduce MIN over the ranks)
    g = cfg['hip_graph']
    assert g['mode'] == 'auto' and 'captured_on_all_ranks' in g and isinstance(g['used'], bool)
    if g['captured_on_all_ranks']:
        assert set(g['trial_ms_per_step']) == {'eager', 'graph'}
    # rank r holds the ids r, r + 2, r + 4, ... (ds_sampler.py:67): disjoint, together 0..7
    assert sorted(int(i) for i in dig2[0]) == [0, 2, 4, 6] and sorted(int(i) for i in dig2[1]) == [1, 3, 5, 7]
    # the same sample gives the same uint8 image whichever rank / batch position / launcher processed it (up to truncation flips: _same_image)
    line1, dig1, img1 = _bench(tmp_path, 'one', 1)
    assert line1['n_gpus'] == 1 and line1['config']['ranks_all_reduced'] == 1 and not line1['config']['ranks_share_devices']
    assert sorted(int(i) for i in dig1[0]) == [0, 1, 2, 3]
    for i in ('0', '2'):
        assert _same_image(img2[0][i], img1[0][i]), f'sample {i}: rank 0 of the two-rank run differs from the single-process run'
    for i in ('1', '3'):
        assert _same_image(img2[1][i], img1[0][i]), f'sample {i}: rank 1 of the two-rank run differs from the single-process run'


def test_bench_eight_ranks_oversubscribed_on_one_device(tmp_path):
    """The launch shape the driver's 8-GPU scaling run uses (``--gpus 8``), oversubscribed on the one device of the test box (256^2, batch 2
    per rank, gloo): eight spawned ranks rendezvous, the all-reduced rank count is 8, rank r holds the ids r, r + 8 (ds_sampler.py:67), the
    HIP-graph decision is common to all ranks, every rank reports its own step and host-enqueue time, and -- with ``--eval-loop`` -- the
    evaluation loop's end-of-run collectives (all-gather of the uint8 results, all-reduce of the FID moments) span the eight ranks."""
    common = [c for c in COMMON if c != '--no-eval-loop']
    common[common.index('--batch') + 1] = '2'
    line, dig, _ = _bench(tmp_path, 'eight', 8, common + ['--eval-loop'])
    cfg = line['config']
    assert line['n_gpus'] == 8 and cfg['global_batch'] == 16 and cfg['launcher'] == 'self-spawn'
    assert cfg['ranks_all_reduced'] == 8 and cfg['ranks_share_devices'] and cfg['collective_backend'] == 'gloo'
    for key in ('ms_per_step_by_rank', 'host_enqueue_ms_per_step_by_rank'):
        assert len(cfg[key]['all']) == 8 and all(v > 0 for v in cfg[key]['all']), key
    g = cfg['hip_graph']
    assert 'captured_on_all_ranks' in g and isinstance(g['used'], bool)
    for r in range(8):
        assert sorted(int(i) for i in dig[r]) == [r, r + 8], (r, dig[r])
    ev = line['eval_loop']
    assert ev['n_gpus'] == 8 and ev['result_ok'] and ev['fid_samples_counted'] == 8 * 2 * 2, ev


def test_bench_two_ranks_training_blocks(tmp_path):
    """``--gpus 2 --all-blocks``: the config-5 blocks with an active gradient all-reduce (gloo here, two ranks on the one device).  Guards
    two things round 6 found on this very command: the instrumented iteration must run on EVERY rank (rank 0 alone stood in its all-reduce
    while rank 1 waited in the next barrier: the run hung until its timeout), and the HIP-graph form with more than one rank -- two graphs
    per phase around the host-side bucket all-reduce -- is captured, timed beside the eager loop and the faster one reported."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--all-blocks', '--train-steps', '1', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--no-second-config', '--no-eval-loop', '--profile-steps', '0', '--watchdog', '600'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    line = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['ranks_all_reduced'] == 2
    for blk in (line['train_step'], line['train_step']['fp16_blocks']):
        assert blk['n_gpus'] == 2 and blk['grad_all_reduce'] == 'gloo' and blk['losses_finite']
        g = blk['hip_graph']
        assert g.get('split_around_all_reduce') is True and g['eager_ms_per_step'] > 0 and g['graph_ms_per_step'] > 0, g
        assert blk['ms_per_step'] == pytest.approx(min(g['eager_ms_per_step'], g['graph_ms_per_step']), rel=1e-3)
        assert blk['kernel_classes_one_step_rank0']
