"""Dataset side of the hot path: FFHQ zip listing, PNG decode, the CoModGAN mask formatter, and the hand-off
of host batches to the device input of the generator.

Mirrors ``lib/data_factory/ds_ffhq.py``:
  * ``ffhqzip_list``          -- ``ffhqzip.init_load_info`` :262-305: the PNG members of ``ffhq{256x256,512x512}.zip`` sorted by
    unique id, ``simple_split`` [0, 10000) = validation, [10000, 70000) = training;
  * ``ZipLoader``             -- :307-330: one open ``ZipFile`` per worker, PNG bytes -> uint8 HWC -> float CHW / 255
    (``torchvision.transforms.ToTensor``).  The reference decodes with ``pyspng``; Pillow yields the same pixels;
  * ``RandomMaskFormatter``   -- :332-347: ``x = image * 2 - 1``, horizontal flip with probability 1/2, ``RandomMask`` -- in this order of
    ``numpy.random`` draws, so a seeded worker produces the reference's sample stream;
  * ``FFHQZip``               -- the ``ds_base.__getitem__`` composition (element -> loader -> formatter);
  * ``DeviceFeeder``          -- what ``shgan_default.py:267-274`` does per batch (``x = cat([mask - 0.5, real * mask])``), on the device:
    pinned staging, H2D on a copy stream overlapped with the previous batch's kernels, masks either from the host formatter or
    drawn on the device (``masks.random_masks``, bit-identical to the host implementation), ``assemble_input`` kernel."""
import io
import os
from zipfile import ZipFile

import numpy as np
import numpy.random as npr
import torch

from . import data as _data

_MODES = {
    'train256': ('ffhq256x256.zip', (10000, 70000)), 'val256': ('ffhq256x256.zip', (0, 10000)),
    'train512': ('ffhq512x512.zip', (10000, 70000)), 'train512ori': ('ffhq512x512.zip', (10000, 70000)),
    'val512': ('ffhq512x512.zip', (0, 10000)), 'val512ori': ('ffhq512x512.zip', (0, 10000)),
}


def ffhqzip_list(root_dir, mode):
    """load_info of ``ffhqzip`` (ds_ffhq.py:262-305)."""
    if mode not in _MODES:
        raise ValueError(mode)
    zipname, split = _MODES[mode]
    zpath = os.path.join(root_dir, zipname)
    info = []
    with ZipFile(zpath, 'r') as z:
        for fi in z.namelist():
            if fi.find('.png') == -1:
                continue
            filename = os.path.basename(fi)
            info.append({'unique_id': os.path.splitext(filename)[0], 'filename': filename, 'image_path': fi, 'zipfile': zpath})
    info = sorted(info, key=lambda x: x['unique_id'])
    info = info[split[0]:split[1]]
    for idx, e in enumerate(info):
        e['idx'] = idx
    return info


class ZipLoader:
    """ds_ffhq.py:307-330: element -> element with 'image' (float32 [C,H,W] in [0,1]) and 'imsize'."""

    def __init__(self):
        self.zipfile, self.zipfilename = None, None

    def decode_u8(self, element):
        """uint8 [H,W,C] of the element's PNG."""
        from PIL import Image
        if self.zipfilename != element['zipfile']:
            self.zipfile_close()
            self.zipfile, self.zipfilename = ZipFile(element['zipfile'], 'r'), element['zipfile']
        with self.zipfile.open(element['image_path'], 'r') as f:
            img = np.asarray(Image.open(io.BytesIO(f.read())))
        return img[:, :, None] if img.ndim == 2 else img

    def __call__(self, element):
        element = dict(element)
        u8 = self.decode_u8(element)
        element['image'] = torch.from_numpy(np.ascontiguousarray(u8.transpose(2, 0, 1))).to(torch.float32).div(255)   # ToTensor
        element['imsize'] = [int(u8.shape[0]), int(u8.shape[1])]
        return element

    def zipfile_close(self):
        if self.zipfile is not None:
            self.zipfile.close()
        self.zipfile, self.zipfilename = None, None


class ImageOnlyFormatter:
    """ds_ffhq.py:247-256."""

    def __init__(self, random_flip=False):
        self.random_flip = random_flip

    def __call__(self, element):
        x = element['image'] * 2 - 1
        if self.random_flip and npr.rand() < 0.5:
            x = x.flip(-1)
        return x, element['unique_id']


class RandomMaskFormatter:
    """ds_ffhq.py:332-347: (x in [-1,1], mask [s,s] float32 with 1 = known -- ``RandomMask(...)[0]``; the eval loop adds the channel axis, shgan_default.py:270 --, unique_id), s = ``mask_resolution`` which must equal the image resolution; the flip draw comes before the mask's."""

    def __init__(self, random_flip=True, mask_resolution=256, hole_range=(0, 1)):
        self.random_flip, self.mask_resolution, self.hole_range = random_flip, mask_resolution, list(hole_range)

    def __call__(self, element):
        x = element['image'] * 2 - 1
        if self.random_flip and npr.rand() < 0.5:
            x = x.flip(-1)
        mask = _data.RandomMask(self.mask_resolution, self.hole_range)[0]
        return x, mask, element['unique_id']


class FFHQZip(torch.utils.data.Dataset):
    """``ffhqzip`` + ``ZipLoader`` + a formatter (ds_base.__getitem__ without its cache / transform options)."""

    def __init__(self, root_dir, mode, formatter=None, try_sample=None, repeat=1):
        self.load_info = ffhqzip_list(root_dir, mode)
        if try_sample is not None:
            self.load_info = self.load_info[:try_sample]
        self.loader, self.formatter, self.repeat = ZipLoader(), formatter, repeat

    def __len__(self):
        return len(self.load_info) * self.repeat

    def __getitem__(self, idx):
        element = self.loader(self.load_info[idx % len(self.load_info)])
        return element if self.formatter is None else self.formatter(element)


class DeviceFeeder:
    """Host batches -> generator inputs on the device.

        feeder = DeviceFeeder(device, resolution=512, device_masks=True)
        for x4, real, mask, ids in feeder(loader):      # loader yields (x [B,3,R,R] in [-1,1], mask [B,1,R,R] or None, ids)
            img = G(x=x4, z=..., c=...)

    The H2D copies of batch k+1 run on a copy stream while batch k's kernels execute (pinned staging buffers, one event per batch);
    ``device_masks`` draws the freeform masks on the device instead of taking the formatter's (same distribution and, for the same
    numpy RNG state, the same bits: ``masks.random_masks``)."""

    def __init__(self, device, resolution, hole_range=(0, 1), device_masks=False, own_stream=True):
        """``own_stream=False``: stage on the CALLER's stream instead of a copy stream -- for callers whose own stream carries nothing
        but this staging (EvalLoop: the generator runs on three side streams).  The copies still overlap with the generator (pinned
        source, asynchronous), the process stays within four HIP streams = the default number of hardware queues, and the mask
        rasteriser's hole-count read never queues behind a generator stream that happens to share a hardware queue with the copy
        stream (measured: a full pipeline drain every third batch, MEASUREMENTS.md round 6)."""
        self.device = torch.device(device)
        self.resolution, self.hole_range, self.device_masks = resolution, tuple(hole_range), device_masks
        if self.device.type == 'cuda' and not own_stream:
            self.copy_stream = None
            self._inline = True
        elif self.device.type == 'cuda':
            from .eval_harness import shared_streams
            self.copy_stream = shared_streams(self.device, 1, kind='copy')[0]       # one staging stream per process (see shared_streams)
        else:
            self.copy_stream = None

    def _to_device(self, t):
        if self.copy_stream is None and getattr(self, '_inline', False):
            return (t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True)
        if self.copy_stream is None:
            return t.to(self.device)
        if not t.is_pinned():
            t = t.pin_memory()
        with torch.cuda.stream(self.copy_stream):
            d = t.to(self.device, non_blocking=True)
        return d

    def _stage(self, batch):
        x, mask, ids = (batch[0], batch[1], batch[2]) if len(batch) == 3 else (batch[0], None, batch[1])
        xd = self._to_device(x.contiguous())
        md = None
        if not self.device_masks:
            if mask is None:
                raise ValueError('DeviceFeeder: the loader yields no masks and device_masks is off')
            m = torch.as_tensor(np.asarray(mask), dtype=torch.float32)
            if m.ndim == 4 and m.shape[1] == 1:
                m = m[:, 0]
            if tuple(m.shape) != (x.shape[0], x.shape[2], x.shape[3]):
                raise ValueError(f'DeviceFeeder: masks {tuple(m.shape)} do not match the images {tuple(x.shape)} -- the formatter\'s '
                                 f'mask_resolution must equal the image resolution ({x.shape[2]}x{x.shape[3]})')
            md = self._to_device(m[:, None].contiguous())
        ev = None
        if self.copy_stream is not None:
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return xd, md, ids, ev

    def _finish(self, staged):
        from . import eval_harness, masks as _masks
        xd, md, ids, ev = staged
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            xd.record_stream(cur)
            if md is not None:
                md.record_stream(cur)
        if md is None:
            # on the copy stream: the rasteriser's hole-count read makes the host wait for the stream it runs on, and this one carries
            # input staging only (on the caller's stream the read waited behind whatever else the caller had queued there)
            if self.copy_stream is not None:
                with torch.cuda.stream(self.copy_stream):
                    md = _masks.random_masks(xd.shape[0], self.resolution, hole_range=self.hole_range, device=self.device).to(torch.float32)
                    md = md.reshape(xd.shape[0], 1, self.resolution, self.resolution)
                cur = torch.cuda.current_stream(self.device)
                cur.wait_stream(self.copy_stream)
                md.record_stream(cur)
            else:
                md = _masks.random_masks(xd.shape[0], self.resolution, hole_range=self.hole_range, device=self.device).to(torch.float32)
                md = md.reshape(xd.shape[0], 1, self.resolution, self.resolution)
        x4 = eval_harness.assemble_input(xd, md)
        return x4, xd, md, ids

    def __call__(self, loader):
        staged = None
        for batch in loader:
            nxt = self._stage(batch)              # copies of batch k+1 are in flight ...
            if staged is not None:
                yield self._finish(staged)        # ... while batch k is handed to the generator
            staged = nxt
        if staged is not None:
            yield self._finish(staged)
