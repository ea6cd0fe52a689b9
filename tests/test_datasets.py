"""Dataset side (shgan_amd.datasets = ds_ffhq.py:247-347 + shgan_default.py:267-274): zip listing and split, PNG decode, the flip /
mask draw order of the formatter, the device hand-off."""
import io
import os
import zipfile

import numpy as np
import numpy.random as npr
import pytest
import torch

import shgan_amd  # noqa: F401
from shgan_amd import data, datasets


def _make_zip(tmp, name, n, res, first_id=0):
    from PIL import Image
    rs = np.random.RandomState(3)
    imgs = {}
    path = os.path.join(tmp, name)
    order = list(range(first_id, first_id + n))
    rs.shuffle(order)                                       # member order in the archive is not the id order
    with zipfile.ZipFile(path, 'w') as z:
        z.writestr('dataset.json', '{}')
        for i in order:
            a = rs.randint(0, 256, (res, res, 3), dtype=np.uint8)
            buf = io.BytesIO()
            Image.fromarray(a).save(buf, format='PNG')
            z.writestr(f'{i // 1000:05d}/img{i:08d}.png', buf.getvalue())
            imgs[f'img{i:08d}'] = a
    return imgs


def test_zip_listing_split_and_decode(tmp_path):
    imgs = _make_zip(str(tmp_path), 'ffhq256x256.zip', 12, 16)
    val = datasets.ffhqzip_list(str(tmp_path), 'val256')
    assert [e['unique_id'] for e in val] == sorted(imgs) and [e['idx'] for e in val] == list(range(12))
    assert datasets.ffhqzip_list(str(tmp_path), 'train256') == []          # ids 10000.. are the training split
    with pytest.raises(ValueError):
        datasets.ffhqzip_list(str(tmp_path), 'train1024')
    ld = datasets.ZipLoader()
    for e in val[:5]:
        out = ld(e)
        assert out['imsize'] == [16, 16] and out['image'].dtype == torch.float32 and tuple(out['image'].shape) == (3, 16, 16)
        assert torch.equal(out['image'], torch.from_numpy(imgs[e['unique_id']].transpose(2, 0, 1).astype(np.float32) / 255))
    ld.zipfile_close()
    assert ld.zipfile is None


def test_formatter_draw_order_and_dataset(tmp_path):
    imgs = _make_zip(str(tmp_path), 'ffhq512x512.zip', 6, 32)
    ds = datasets.FFHQZip(str(tmp_path), 'val512', formatter=datasets.RandomMaskFormatter(True, 32, [0, 1]))
    assert len(ds) == 6
    npr.seed(11)
    got = [ds[i] for i in range(6)]
    npr.seed(11)                                             # the same stream by hand: flip draw, then the mask's draws (ds_ffhq.py:343-346)
    for i, (x, mask, uid) in enumerate(got):
        ref = torch.from_numpy(imgs[uid].transpose(2, 0, 1).astype(np.float32) / 255) * 2 - 1
        if npr.rand() < 0.5:
            ref = ref.flip(-1)
        m = data.RandomMask(32, [0, 1])[0]
        assert uid == sorted(imgs)[i] and torch.equal(x, ref) and np.array_equal(mask, m) and mask.shape == (32, 32)
    x, uid = datasets.FFHQZip(str(tmp_path), 'val512', formatter=datasets.ImageOnlyFormatter(False))[2]
    assert float(x.min()) >= -1 and float(x.max()) <= 1 and uid == sorted(imgs)[2]


def test_device_feeder_on_the_host_device(tmp_path):
    _make_zip(str(tmp_path), 'ffhq256x256.zip', 8, 16)
    ds = datasets.FFHQZip(str(tmp_path), 'val256', formatter=datasets.RandomMaskFormatter(True, 16, [0, 1]))
    npr.seed(5)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=0)
    feeder = datasets.DeviceFeeder('cpu', 16)
    seen = 0
    for x4, real, mask, ids in feeder(loader):
        assert tuple(x4.shape) == (real.shape[0], 4, 16, 16) and tuple(mask.shape) == (real.shape[0], 1, 16, 16)
        assert torch.equal(x4, torch.cat([mask - 0.5, real * mask], dim=1)) and set(np.unique(mask.numpy())) <= {0.0, 1.0}
        seen += real.shape[0]
    assert seen == 8


@pytest.mark.gpu
def test_device_feeder_on_the_gpu_host_and_device_masks(tmp_path):
    """Pinned staging + copy stream + assemble_input kernel; device-drawn masks are the host formatter's masks bit for bit when the
    numpy RNG is in the same state."""
    from shgan_amd import masks
    _make_zip(str(tmp_path), 'ffhq256x256.zip', 10, 64)
    ds = datasets.FFHQZip(str(tmp_path), 'val256', formatter=datasets.RandomMaskFormatter(False, 64, [0, 1]))
    npr.seed(9)
    host = [(x4.cpu(), m.cpu(), ids) for x4, _, m, ids in
            datasets.DeviceFeeder('cuda:0', 64)(torch.utils.data.DataLoader(ds, batch_size=4, num_workers=0))]
    for x4, m, _ in host:
        assert x4.shape[1] == 4 and torch.equal(x4[:, :1], m - 0.5)
    # the same images with masks drawn on the device: same RNG state -> same masks as the host formatter drew
    ds_img = datasets.FFHQZip(str(tmp_path), 'val256', formatter=datasets.ImageOnlyFormatter(False))
    npr.seed(9)
    dev = list(datasets.DeviceFeeder('cuda:0', 64, device_masks=True)(torch.utils.data.DataLoader(ds_img, batch_size=4, num_workers=0)))
    assert len(dev) == len(host) == 3
    for (x4h, mh, idh), (x4d, _, md, idd) in zip(host, dev):
        assert list(idh) == list(idd) and torch.equal(md.cpu(), mh) and torch.equal(x4d.cpu(), x4h)
