"""FID statistics (reference lib/evaluator/eva_fid.py:194-277, eva_base.py:96-230) on synthetic
[B, 2048] features (the Inception detector is a download and cannot be pinned offline).  The checker is the REFERENCE: ``tests/golden/fid.npz``
holds FID values computed by the reference's own ``zipzap_arrange`` + ``compute_fid`` (tools/gen_golden.py: gen_fid) on feature sets sharded by
its ``DistributedSampler(extend=True)``; ``reference_fid`` below (numpy restatement of eva_fid.py:239-261) is itself held to those values
and serves only for intermediate quantities (mu, sigma) the reference does not expose."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import shgan_amd  # noqa: F401
from conftest import ROOT
from shgan_amd import fid_stats


def reference_fid(fake, real, n_fake=None, n_real=None):
    """eva_fid.py:239-261 on float64 feature matrices in dataset order."""
    import scipy.linalg
    fake, real = fake[:n_fake], real[:n_real]
    mu_f, mu_r = fake.mean(0), real.mean(0)
    sig_f = fake.T @ fake / fake.shape[0] - np.outer(mu_f, mu_f)
    sig_r = real.T @ real / real.shape[0] - np.outer(mu_r, mu_r)
    s, _ = scipy.linalg.sqrtm(np.dot(sig_f, sig_r), disp=False)
    return float(np.real(np.square(mu_f - mu_r).sum() + np.trace(sig_f + sig_r - s * 2))), (mu_f, sig_f, mu_r, sig_r)


def golden_cases():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'fid.npz'))
    for name in z['names']:
        name = str(name)
        D, n, world, bs, sfn, srn = (int(v) for v in z[name + '__meta'])
        yield name, dict(D=D, n=n, world=world, bs=bs, sample_fake_n=n if sfn < 0 else sfn, sample_real_n=n if srn < 0 else srn,
                         real=z[name + '__real'], fake=z[name + '__fake'], shards=z[name + '__shards'], fid=float(z[name + '__fid']))


def replay_sharded(case, device, accumulate_fn):
    """Every rank's batches through ``FidStats.add_shard`` (ranks emulated in turn; summing their moments is what the one all-reduce does)."""
    out = []
    for feats, sample_n in ((case['fake'], case['sample_fake_n']), (case['real'], case['sample_real_n'])):
        total = fid_stats.FidStats(case['D'], device, accumulate_fn)
        for r in range(case['world']):
            st = fid_stats.FidStats(case['D'], device, accumulate_fn)
            ids = case['shards'][r]
            for k0 in range(0, len(ids), case['bs']):
                st.add_shard(torch.from_numpy(feats[ids[k0:k0 + case['bs']]]).to(device), k0, r, case['world'], sample_n=sample_n)
            total.S += st.S
        out.append(total.mean_cov(sample_n))
    return out


@pytest.mark.parametrize('name,case', list(golden_cases()))
def test_fid_statistics_match_the_reference_compute_fid(name, case):
    """Host logic (weights for padded duplicates, truncation to sample_n, moments -> mu/sigma -> sqrtm tail) against the reference's own
    ``compute_fid`` value; the restatement used elsewhere in this file is pinned on the same values."""
    (n_f, mu_f, sig_f), (n_r, mu_r, sig_r) = replay_sharded(case, 'cpu', numpy_accumulate)
    assert (n_f, n_r) == (case['sample_fake_n'], case['sample_real_n'])
    got = fid_stats.fid_from_stats(mu_f, sig_f, mu_r, sig_r)
    assert abs(got - case['fid']) <= 1e-7 * abs(case['fid']), (got, case['fid'])
    ref, _ = reference_fid(case['fake'].astype(np.float64), case['real'].astype(np.float64), case['sample_fake_n'], case['sample_real_n'])
    assert abs(ref - case['fid']) <= 1e-7 * abs(case['fid']), (ref, case['fid'])


def test_empty_accumulator_raises():
    st = fid_stats.FidStats(8, 'cpu', numpy_accumulate)
    with pytest.raises(ValueError):
        st.mean_cov()
    st.add(torch.ones(3, 8), torch.zeros(3))
    with pytest.raises(ValueError):
        st.mean_cov()


def test_add_images_hands_the_batch_to_the_detector():
    """eva_fid.py:194-206: detector(images.float(), return_features=True) on 0..255 images; padded duplicates weighted out."""
    calls = []

    def detector(img, return_features=False):
        calls.append((img.dtype, tuple(img.shape), return_features))
        return img.flatten(1)[:, :8] / 255.0

    st = fid_stats.FidStats(8, 'cpu', numpy_accumulate)
    u8 = torch.arange(4 * 3 * 2 * 2, dtype=torch.uint8).reshape(4, 3, 2, 2)
    f = st.add_images(detector, u8, k0=2, rank=1, world=2, sample_n=10)      # positions 5, 7, 9, 11 -> the last one is padding
    assert calls == [(torch.float32, (4, 3, 2, 2), True)]
    n, mu, _ = st.mean_cov()
    assert n == 3 and np.allclose(mu, f[:3].double().mean(0).numpy())
    with pytest.raises(Exception):
        st.add_images(lambda img, return_features: img.flatten(1)[:, :5], u8)


def numpy_accumulate(S, feats, weights):
    """Stand-in for the HIP kernel on CPU tensors (tests only): S += sum_b w_b [x,1][x,1]^T."""
    x = feats.detach().cpu().numpy().astype(np.float64)
    w = np.ones(len(x)) if weights is None else weights.detach().cpu().numpy().astype(np.float64)
    xa = np.concatenate([x, np.ones((len(x), 1))], axis=1)
    d = xa.shape[1]
    S[:d, :d] += torch.from_numpy((xa * w[:, None]).T @ xa)


def test_fid_from_moments_equals_reference_formula():
    rs = np.random.RandomState(0)
    d = 48
    fake = rs.standard_normal((300, d)) @ rs.standard_normal((d, d)) * 0.3 + 0.2
    real = rs.standard_normal((280, d)) @ rs.standard_normal((d, d)) * 0.3
    ref, (mu_f, sig_f, mu_r, sig_r) = reference_fid(fake, real)
    a, b = fid_stats.FidStats(d, 'cpu', numpy_accumulate), fid_stats.FidStats(d, 'cpu', numpy_accumulate)
    for k in range(0, 300, 37):
        a.add(torch.from_numpy(fake[k:k + 37]))
    b.add(torch.from_numpy(real))
    n_f, m_f, s_f = a.mean_cov()
    n_r, m_r, s_r = b.mean_cov()
    assert (n_f, n_r) == (300, 280)
    assert np.allclose(m_f, mu_f, rtol=0, atol=1e-12) and np.allclose(s_f, sig_f, rtol=0, atol=1e-10)
    assert abs(fid_stats.fid_from_stats(m_f, s_f, m_r, s_r) - ref) < 1e-8 * max(1.0, abs(ref))


def test_gloo_world2_moment_allreduce_and_feature_gather():
    """Two gloo ranks shard an 11-item feature set as DistributedSampler(extend=True) does (12 slots, one padded duplicate):
    the all-reduced moments with the duplicate weighted out, and the gathered + re-interleaved features, both reproduce the
    reference's single-list statistics."""
    script = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SHG_ROOT"], "tests"))
import shgan_amd
from shgan_amd import fid_stats, eval_harness as hz
from test_fid_stats import numpy_accumulate, reference_fid
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"], rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
n, d = 11, 24
rs = np.random.RandomState(4)
fake = rs.standard_normal((n, d)) * 0.5 + 0.1; real = rs.standard_normal((n, d))
ids = hz.shard_ids(n, r, 2)                                   # rank 1 receives the padded duplicate of item 0
stats = []
for feats in (fake, real):
    st = fid_stats.FidStats(d, "cpu", numpy_accumulate)
    for k0 in range(0, len(ids), 4):                          # batches of 4 items
        st.add_shard(torch.from_numpy(feats[ids[k0:k0 + 4]]), k0, r, 2, sample_n=n)
    stats.append(st.all_reduce().mean_cov())
ref, (mu_f, sig_f, mu_r, sig_r) = reference_fid(fake, real)
assert stats[0][0] == n and stats[1][0] == n, (stats[0][0], stats[1][0])
assert np.allclose(stats[0][1], mu_f, atol=1e-12) and np.allclose(stats[1][2], sig_r, atol=1e-10)
got = fid_stats.fid_from_stats(stats[0][1], stats[0][2], stats[1][1], stats[1][2])
assert abs(got - ref) < 1e-8 * max(1.0, abs(ref)), (got, ref)
g = fid_stats.gather_features(torch.from_numpy(fake[ids]), n, r, 2)
assert g.shape == (n, d) and np.array_equal(g, fake)
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


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_device_moment_kernel_vs_numpy(dtype):
    """shg_fid_accumulate_f64 (fp64 MFMA) at the real feature width, ragged batches, padded-duplicate weights."""
    rs = np.random.RandomState(7)
    d = 2048
    st = fid_stats.FidStats(d, 'cuda:0')
    S = np.zeros((d + 1, d + 1))
    for b in (16, 5, 33):
        x = (rs.standard_normal((b, d)) * rs.rand(d) + rs.rand(d)).astype(np.float32 if dtype == torch.float32 else np.float64)
        w = (rs.rand(b) < 0.8).astype(np.float32)
        st.add(torch.from_numpy(x).to('cuda:0'), torch.from_numpy(w).to('cuda:0'))
        xa = np.concatenate([x.astype(np.float64), np.ones((b, 1))], axis=1)
        S += (xa * w[:, None].astype(np.float64)).T @ xa
    got = st.S.cpu().numpy()[:d + 1, :d + 1]
    got = np.triu(got) + np.triu(got, 1).T
    assert np.abs(got - S).max() <= 1e-12 * np.abs(S).max()
    n, mu, sigma = st.mean_cov()
    assert n == S[d, d] and np.allclose(mu, S[:d, d] / n, atol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize('name,case', list(golden_cases()))
def test_device_fid_matches_the_reference_compute_fid(name, case):
    """The HIP moment kernel end to end against the reference's ``compute_fid`` values (tests/golden/fid.npz): sharded replay with padded
    duplicates, world 1 / 2 / 4, feature width 64 and 2048."""
    (n_f, mu_f, sig_f), (n_r, mu_r, sig_r) = replay_sharded(case, 'cuda:0', None)
    assert (n_f, n_r) == (case['sample_fake_n'], case['sample_real_n'])
    got = fid_stats.fid_from_stats(mu_f, sig_f, mu_r, sig_r)
    assert abs(got - case['fid']) <= 1e-6 * abs(case['fid']), (got, case['fid'])


@pytest.mark.gpu
def test_device_fid_end_to_end_vs_reference_formula():
    rs = np.random.RandomState(8)
    d = 256
    mix = rs.standard_normal((d, d)) * 0.2
    fake = (rs.standard_normal((512, d)) @ mix + 0.1).astype(np.float32)
    real = (rs.standard_normal((480, d)) @ mix).astype(np.float32)
    ref, _ = reference_fid(fake.astype(np.float64), real.astype(np.float64))
    a, b = fid_stats.FidStats(d, 'cuda:0'), fid_stats.FidStats(d, 'cuda:0')
    for k in range(0, 512, 16):
        a.add(torch.from_numpy(fake[k:k + 16]).to('cuda:0'))
    for k in range(0, 480, 16):
        b.add(torch.from_numpy(real[k:k + 16]).to('cuda:0'))
    _, mf, sf = a.mean_cov()
    _, mr, sr = b.mean_cov()
    got = fid_stats.fid_from_stats(mf, sf, mr, sr)
    assert abs(got - ref) < 1e-6 * max(1.0, abs(ref)), (got, ref)
