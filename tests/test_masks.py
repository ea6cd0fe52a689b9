"""Freeform-mask synthesis (reference lib/data_factory/ds_ffhq.py:145-217).
CPU: the host half (random draws in the reference's order -> primitive records) and the CPU restatement of the
rasteriser, against Pillow itself (fuzz) and against the reference-generated golden masks.  GPU (-m gpu): the HIP
rasteriser, bit-exact against the same golden masks, the host RandomMask on further seeds, the batched generator
with its rejection / RNG-rewind logic, and the hand-off into the generator input without a host round trip."""
import numpy as np
import pytest
import torch

import shgan_amd  # noqa: F401
from conftest import load_golden
from oracle import mask_raster_oracle as mo
from shgan_amd import data, masks


def golden_masks(s):
    g = load_golden('integer_paths')
    return [np.unpackbits(g[f'mask{s}_bits'][k])[: s * s].reshape(s, s) for k in range(4)]


def host_mask_via_records(s, hole_range=(0, 1)):
    """RandomMask's loop with the record generator + CPU rasteriser in place of Pillow."""
    tab = masks.disc_span_table()
    while True:
        rec, f0, f1 = masks.mask_attempt_records(s, hole_range)
        m = mo.rasterize(rec, f0, f1, s, tab)
        hole = 1 - m.mean()
        if hole <= hole_range[0] or hole >= hole_range[1]:
            continue
        return m


def test_cpu_rasteriser_matches_pillow_on_random_segments_and_discs():
    from PIL import Image, ImageDraw
    rs = np.random.RandomState(3)
    tab = masks.disc_span_table()
    s = 64
    for _ in range(400):
        pts = [(int(v[0]), int(v[1])) for v in rs.randint(-6, s + 6, size=(int(rs.randint(2, 5)), 2))]
        width = int(rs.randint(2, 49))
        ref = Image.new('L', (s, s), 0)
        pen = ImageDraw.Draw(ref)
        pen.line(pts, fill=1, width=width)
        for (vx, vy) in pts:
            pen.ellipse((vx - width // 2, vy - width // 2, vx + width // 2, vy + width // 2), fill=1)
        recs = masks.thick_polyline_records(pts, width, s)
        discs = np.zeros((len(pts), 8), np.int32)
        discs[:, 0], discs[:, 1:3], discs[:, 3] = masks.DISC, np.asarray(pts), width // 2
        got = 1 - mo.rasterize(np.concatenate(recs + [discs]), False, False, s, tab)
        assert np.array_equal(got, np.asarray(ref, np.uint8)), (pts, width)


@pytest.mark.parametrize('s', [64, 256])
def test_host_records_reproduce_reference_masks_bit_exactly(s):
    """numpy global RNG seeded as tools/gen_golden.py did: the record generator draws in the reference's order."""
    np.random.seed(0)
    for k, ref in enumerate(golden_masks(s)):
        assert np.array_equal(host_mask_via_records(s), ref), k
    # the RNG stream afterwards is where the reference's own implementation leaves it
    st = np.random.get_state()[1].copy()
    np.random.seed(0)
    for _ in range(4):
        data.RandomMask(s, [0, 1])
    assert np.array_equal(st, np.random.get_state()[1])


@pytest.mark.gpu
@pytest.mark.parametrize('s', [64, 256, 512])
def test_device_masks_bit_exact_vs_golden_and_host(s):
    np.random.seed(0)
    got = masks.random_masks(4, s, [0, 1], device='cuda:0', batch=3)          # batch 3: the generator crosses a batch border
    st = np.random.get_state()[1].copy()
    for k, ref in enumerate(golden_masks(s)):
        assert np.array_equal(got[k, 0].cpu().numpy().astype(np.uint8), ref), k
    np.random.seed(0)
    for _ in range(4):
        data.RandomMask(s, [0, 1])
    assert np.array_equal(st, np.random.get_state()[1])
    # further seeds against the host implementation (Pillow), incl. a narrow hole range that exercises the rejection loop
    for seed, hr in ((11, [0, 1]), (12, [0.3, 0.5]), (13, [0, 1])):
        np.random.seed(seed)
        ref = np.stack([data.RandomMask(s, hr) for _ in range(5)])
        ref_state = np.random.get_state()[1].copy()
        np.random.seed(seed)
        dev = masks.random_masks(5, s, hr, device='cuda:0', batch=4)
        assert np.array_equal(dev.cpu().numpy(), ref), (seed, hr)
        assert np.array_equal(ref_state, np.random.get_state()[1])


@pytest.mark.gpu
def test_device_masks_feed_the_generator_input_without_host_round_trip():
    from shgan_amd import eval_harness as hz
    np.random.seed(21)
    m = masks.random_masks(3, 256, [0, 1], device='cuda:0')
    assert m.is_cuda and tuple(m.shape) == (3, 1, 256, 256) and set(np.unique(m.cpu().numpy())) <= {0.0, 1.0}
    real = torch.rand(3, 3, 256, 256, device='cuda:0') * 2 - 1
    x = hz.assemble_input(real, m)
    assert torch.equal(x[:, 0:1], m - 0.5) and torch.equal(x[:, 1:4], real * m)
