"""On-device synthesis of the freeform masks of the eval / training loops (reference:
lib/data_factory/ds_ffhq.py:145-217 ``RandomBrush`` / ``RandomMask``).

Split of the work:
  * HOST (this file): the random draws, made from numpy's global ``RandomState`` in exactly the reference's order, and the
    geometry that depends only on them -- every mask becomes a short list of integer primitives (rectangles to punch,
    thick-line quads, discs, the two flip decisions);
  * DEVICE (csrc/mask_raster.hip, ``shg_mask_raster_f32``): rasterisation of the primitives, the AND of the two layers, the
    hole count, and -- through ``shg_assemble_input_f32`` -- the generator input ``cat([mask-0.5, real*mask])`` without a
    host round trip.

Bit-exactness: the reference draws with Pillow (``ImageDraw.line(width=..)`` + ``ImageDraw.ellipse``), so Pillow's
rasteriser is the specification.  The quad of a thick segment and its scan-line fill are restated from Pillow's
``ImagingDrawWideLine`` / ``polygon_generic`` (float32 edge slopes, the ROUND_UP / ROUND_DOWN span rule, the corner
joining); disc spans are read once per radius from Pillow itself (translation invariant, 18 radii).  The golden masks
of tests/golden/integer_paths.npz pin the result.

The rejection loop of ``RandomMask`` (re-draw while the hole ratio is outside ``hole_range``) needs the rasterised mask:
masks are generated speculatively in batches, the device returns the hole counts, and on a rejection the global RNG is
rewound to the state after the rejected attempt and the rest of the batch is drawn again -- the RNG stream, and so the
masks, equal the reference's sequential loop."""
import math

import numpy as np
import torch

from . import _lib, kernels
from ._lib import check

RECT, DISC, QUAD, EDGE, POINT, SEG = 0, 1, 2, 3, 4, 5
REC = 8                                   # int32 words per primitive record
MAX_HALF = 32                             # disc radii served by the span table


def _round_up(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= 0.0, np.floor(f + 0.5), -np.floor(np.abs(f) + 0.5)).astype(np.int64)


def _round_down(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= 0.0, np.ceil(f - 0.5), -np.ceil(np.abs(f) - 0.5)).astype(np.int64)


def _edge_records(xa, ya, xb, yb):
    """[n] edges (xa,ya)->(xb,yb) -> [n, 8] EDGE records (Pillow ``add_edge``): slope in float32."""
    n = len(xa)
    rec = np.zeros((n, REC), dtype=np.int32)
    rec[:, 0] = EDGE
    rec[:, 1], rec[:, 2] = xa, ya
    rec[:, 3], rec[:, 4] = np.minimum(ya, yb), np.maximum(ya, yb)
    dy = (yb - ya).astype(np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        dx = np.where(dy != 0, (xb - xa).astype(np.float32) / np.where(dy != 0, dy, np.float32(1)), np.float32(0)).astype(np.float32)
    rec[:, 5] = dx.view(np.int32)
    rec[:, 6], rec[:, 7] = np.minimum(xa, xb), np.maximum(xa, xb)
    return rec


def thick_polyline_records(path, width, s):
    """Records of ``ImageDraw.line(path, width=width)`` on an s x s canvas: one QUAD (+ 4 EDGE) per segment
    (``ImagingDrawWideLine``), a POINT for a zero-length segment."""
    p = np.asarray(path, dtype=np.int64)
    x0, y0, x1, y1 = p[:-1, 0], p[:-1, 1], p[1:, 0], p[1:, 1]
    dx, dy = x1 - x0, y1 - y0
    out = []
    zero = (dx == 0) & (dy == 0)
    big = np.hypot(dx.astype(np.float64), dy.astype(np.float64))
    big = np.where(zero, 1.0, big)
    small = (width - 1) / 2.0
    rmax, rmin = float(_round_up(small)) / big, float(_round_down(small)) / big
    dxmin, dxmax = _round_down(rmin * dy), _round_down(rmax * dy)
    dymin, dymax = _round_up(rmin * dx), _round_up(rmax * dx)
    vx = np.stack([x0 - dxmin, x1 - dxmin, x1 + dxmax, x0 + dxmax], axis=1)      # [n, 4]
    vy = np.stack([y0 + dymax, y1 + dymax, y1 - dymin, y0 - dymin], axis=1)
    nseg = len(x0)
    e = _edge_records(vx.reshape(-1), vy.reshape(-1), np.roll(vx, -1, axis=1).reshape(-1), np.roll(vy, -1, axis=1).reshape(-1))
    e = e.reshape(nseg, 4, REC)
    head = np.zeros((nseg, 1, REC), dtype=np.int32)
    head[:, 0, 0] = QUAD
    head[:, 0, 1] = np.maximum(vy.min(axis=1), 0)                 # polygon_generic clamps the scan range to [0, ysize]
    head[:, 0, 2] = np.minimum(vy.max(axis=1), s)
    quads = np.concatenate([head, e], axis=1)                     # [n, 5, 8]
    for k in range(nseg):
        if zero[k]:
            pt = np.zeros((1, REC), dtype=np.int32)
            pt[0, :3] = (POINT, x0[k], y0[k])
            out.append(pt)
        else:
            out.append(quads[k])
    return out


def segment_records(path, width):
    """Records of ``ImageDraw.line(path, width=width)`` with the quad geometry left to the device: one SEG per segment,
    carrying hypot(dx, dy) as computed by the host's libm (the one input of Pillow's quad that is not exact arithmetic)."""
    p = np.asarray(path, dtype=np.int32)
    n = len(p) - 1
    rec = np.zeros((n, REC), dtype=np.int32)
    rec[:, 0] = SEG
    rec[:, 1:3], rec[:, 3:5] = p[:-1], p[1:]
    rec[:, 5] = width
    d = (p[1:] - p[:-1]).astype(np.float64)
    rec[:, 6:8] = np.hypot(d[:, 0], d[:, 1]).view(np.int32).reshape(n, 2)          # little endian: lo, hi
    return rec


def brush_records(max_tries, s, min_num_vertex=4, max_num_vertex=18, mean_angle=2 * math.pi / 5, angle_range=2 * math.pi / 15,
                  min_width=12, max_width=48):
    """The draws of ``RandomBrush`` (ds_ffhq.py:145-197) in the reference's order -> (records [n,8], flip0, flip1)."""
    rng = np.random
    mean_radius = math.sqrt(s * s + s * s) / 8
    recs = []
    for _ in range(rng.randint(max_tries)):
        n_vertex = rng.randint(min_num_vertex, max_num_vertex)
        lo = mean_angle - rng.uniform(0, angle_range)
        hi = mean_angle + rng.uniform(0, angle_range)
        ang = rng.uniform(lo, hi, size=n_vertex)                  # same stream as n_vertex scalar draws
        ang[0::2] = 2 * math.pi - ang[0::2]
        path = [(int(rng.randint(0, s)), int(rng.randint(0, s)))]
        steps = np.clip(rng.normal(loc=mean_radius, scale=mean_radius // 2, size=n_vertex), 0, 2 * mean_radius)
        for a, st in zip(ang.tolist(), steps.tolist()):
            px = min(max(path[-1][0] + st * math.cos(a), 0), s)
            py = min(max(path[-1][1] + st * math.sin(a), 0), s)
            path.append((int(px), int(py)))
        thick = int(rng.uniform(min_width, max_width))
        recs.append(segment_records(path, thick))
        discs = np.zeros((len(path), REC), dtype=np.int32)
        discs[:, 0] = DISC
        discs[:, 1:3] = np.asarray(path, dtype=np.int32)
        discs[:, 3] = thick // 2
        recs.append(discs)
        rng.random()                                              # two flip decisions the reference draws and discards
        rng.random()
    flip0 = bool(rng.random() > 0.5)
    flip1 = bool(rng.random() > 0.5)
    return recs, flip0, flip1


def mask_attempt_records(s, hole_range=(0, 1)):
    """One pass of the ``while True`` body of ``RandomMask`` (ds_ffhq.py:199-217) -> (records [n,8] int32, flip0, flip1)."""
    rng = np.random
    coef = min(hole_range[0] + hole_range[1], 1.0)
    recs = []
    for max_tries, max_size in ((int(10 * coef), s // 2), (int(5 * coef), s)):
        for _ in range(rng.randint(max_tries)):
            w, h = rng.randint(max_size), rng.randint(max_size)
            x, y = rng.randint(-(w // 2), s - w + w // 2), rng.randint(-(h // 2), s - h + h // 2)
            x0, x1, y0, y1 = max(x, 0), min(x + w, s) - 1, max(y, 0), min(y + h, s) - 1
            if x0 <= x1 and y0 <= y1:
                r = np.zeros((1, REC), dtype=np.int32)
                r[0, :5] = (RECT, x0, x1, y0, y1)
                recs.append(r)
    brush, f0, f1 = brush_records(int(20 * coef), s)
    recs.extend(brush)
    if recs:
        return np.concatenate(recs, axis=0), f0, f1
    return np.zeros((0, REC), dtype=np.int32), f0, f1


_disc_table = None


def disc_span_table():
    """[MAX_HALF+1, 2*MAX_HALF+1, 2] int32: row j of the filled ellipse with bounding box (c-h, c-h, c+h, c+h) covers columns
    c-h+l .. c-h+r (l > r: empty row).  Read from Pillow -- the reference's rasteriser is the specification."""
    global _disc_table
    if _disc_table is None:
        from PIL import Image, ImageDraw
        t = np.zeros((MAX_HALF + 1, 2 * MAX_HALF + 1, 2), dtype=np.int32)
        t[:, :, 0] = 1
        for h in range(1, MAX_HALF + 1):
            c = Image.new('L', (2 * h + 9, 2 * h + 9), 0)
            ImageDraw.Draw(c).ellipse((4, 4, 4 + 2 * h, 4 + 2 * h), fill=1)
            a = np.asarray(c)
            for j in range(2 * h + 1):
                xs = np.nonzero(a[4 + j])[0]
                if len(xs):
                    t[h, j] = (int(xs[0]) - 4, int(xs[-1]) - 4)
        _disc_table = t
    return _disc_table


def rasterize(records, offsets, flips, s, device='cuda'):
    """records [total,8] int32, offsets [B+1], flips [B,2] (host arrays) -> (mask float32 [B,1,s,s] with 1 = keep / 0 = hole,
    hole counts int32 [B]) on ``device``: one H2D copy of the primitive lists, one kernel."""
    if s > 512 or s % 32 != 0:
        raise _lib.ShgError('mask rasteriser: s must be a multiple of 32, at most 512')
    if records.shape[0] and int(records[records[:, 0] == DISC, 3].max(initial=0)) > MAX_HALF:
        raise _lib.ShgError('mask rasteriser: disc radius beyond the span table')
    dev = torch.device(device)
    b = len(offsets) - 1
    # ONE H2D copy of the primitive lists, from pinned memory: a pageable source of this size (~0.5 MB) takes the runtime's
    # pin-in-place path, which was measured to stall the host for two batches of device work every fourth batch of the evaluation
    # loop (37 ms; MEASUREMENTS.md, round 6).  torch's caching host allocator keeps the block alive until the copy has run.
    rec_h = np.ascontiguousarray(records.reshape(-1), dtype=np.int32) if records.shape[0] else np.zeros(REC, np.int32)
    off_h = np.asarray(offsets, dtype=np.int32)
    flip_h = np.asarray(flips, dtype=np.int32).reshape(-1)
    n_rec, n_off, n_flip = rec_h.size, off_h.size, flip_h.size
    o_off = (n_rec + 3) // 4 * 4                         # 16-byte aligned sections
    o_flip = o_off + (n_off + 3) // 4 * 4
    stage = torch.empty(o_flip + n_flip, dtype=torch.int32, pin_memory=(dev.type == 'cuda'))
    sv = stage.numpy()
    sv[:n_rec], sv[o_off:o_off + n_off], sv[o_flip:o_flip + n_flip] = rec_h, off_h, flip_h
    stage_d = stage.to(dev, non_blocking=True)
    rec_d, off_d, flip_d = stage_d[:n_rec], stage_d[o_off:o_off + n_off], stage_d[o_flip:o_flip + n_flip]
    tab_d = _table_on(dev)
    mask = torch.empty((b, 1, s, s), dtype=torch.float32, device=dev)
    holes = torch.zeros((b,), dtype=torch.int32, device=dev)
    L = kernels._Launch()
    for t, nm in ((rec_d, 'records'), (off_d, 'offsets'), (flip_d, 'flips'), (tab_d, 'table'), (holes, 'holes')):
        L.req(t, nm, dtype=torch.int32)
    L.req(mask, 'mask')
    with L:
        check(_lib.get_lib().shg_mask_raster_f32(kernels._ptr(rec_d), kernels._ptr(off_d), kernels._ptr(flip_d), kernels._ptr(tab_d),
                                                 MAX_HALF, kernels._ptr(mask), kernels._ptr(holes), b, s, L.stream()), 'mask_raster')
    return mask, holes


_tables = {}


def _table_on(dev):
    key = str(dev)
    if key not in _tables:
        _tables[key] = torch.from_numpy(disc_span_table().reshape(-1).copy()).to(dev)
    return _tables[key]


def random_masks(n, s, hole_range=(0, 1), device='cuda', batch=64):
    """``n`` masks of ``RandomMask(s, hole_range)`` drawn from numpy's global RNG, rasterised on ``device``:
    float32 [n,1,s,s].  Same masks (and same final RNG state) as n sequential calls of the reference function."""
    out = []
    while len(out) < n:
        want = min(batch, n - len(out))
        states, recs, offs, flips = [], [], [0], []
        for _ in range(want):
            r, f0, f1 = mask_attempt_records(s, hole_range)
            states.append(np.random.get_state())
            recs.append(r)
            offs.append(offs[-1] + len(r))
            flips.append((int(f0), int(f1)))
        mask, holes = rasterize(np.concatenate(recs, axis=0) if offs[-1] else np.zeros((0, REC), np.int32), offs, flips, s, device)
        ratio = holes.cpu().numpy().astype(np.float64) / float(s * s)        # the one synchronisation per batch
        ok = ~((ratio <= hole_range[0]) | (ratio >= hole_range[1])) if hole_range is not None else np.ones(want, bool)
        if ok.all():
            out.extend(mask[k] for k in range(want))
            continue
        bad = int(np.argmin(ok))                       # first rejected attempt: everything after it was drawn from a wrong state
        out.extend(mask[k] for k in range(bad))
        np.random.set_state(states[bad])               # the reference loops: the next attempt continues from here
    return torch.stack(out[:n]) if out else torch.empty((0, 1, s, s), device=device)
