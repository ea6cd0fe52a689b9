"""TEST INFRASTRUCTURE (never imported by the product): CPU restatement of the primitive rasteriser of
sh-gan_amd/csrc/mask_raster.hip, i.e. of the parts of Pillow that lib/data_factory/ds_ffhq.py:145-197 draws with
(``ImageDraw.line(width)`` = ``ImagingDrawWideLine`` -> ``polygon_generic``; ``ImageDraw.ellipse(fill)``).
Pinned against Pillow itself (tests/test_masks.py fuzzes random segments / discs) and against the reference-generated
golden masks of tests/golden/integer_paths.npz.  Pure-Python loops: small cases only."""
import math

import numpy as np

f32 = np.float32
RECT, DISC, QUAD, EDGE, POINT, SEG = 0, 1, 2, 3, 4, 5


def round_up(f):
    return int(math.floor(f + 0.5)) if f >= 0.0 else -int(math.floor(abs(f) + 0.5))


def round_down(f):
    return int(math.ceil(f - 0.5)) if f >= 0.0 else -int(math.ceil(abs(f) - 0.5))


def _hline(im, x0, y, x1):
    h, w = im.shape
    if 0 <= y < h:
        if x0 < 0:
            x0 = 0
        elif x0 >= w:
            return
        if x1 < 0:
            return
        if x1 >= w:
            x1 = w - 1
        if x0 <= x1:
            im[y, x0:x1 + 1] = 1


def _ex(e, y):
    """(y - y0) * dx + x0 evaluated in float32, product and sum rounded separately (no FMA)."""
    return f32(f32(y - int(e[2])) * e[5]) + f32(int(e[1]))


def fill_quad(im, head, edges):
    """Pillow's polygon_generic on the four edges of a thick-line quad; edges: rows (type, x0, y0, ymin, ymax, dx, xmin, xmax)."""
    h = im.shape[0]
    table = []
    for e in edges:
        if e[3] == e[4]:
            _hline(im, int(e[6]), int(e[3]), int(e[7]))
        else:
            table.append(e)
    ymin, ymax = int(head[1]), int(head[2])
    for y in range(ymin, min(ymax, h) + 1):
        xx = []
        for i, cur in enumerate(table):
            if cur[3] <= y <= cur[4]:
                xx.append(_ex(cur, y))
                j = len(xx)
                if y == cur[4] and y < ymax:
                    xx.append(xx[-1])
                elif cur[5] != 0 and j % 2 == 1 and f32(np.round(xx[j - 1])) == xx[j - 1]:
                    for k in range(i):
                        oth = table[k]
                        if (cur[5] > 0 and oth[5] <= 0) or (cur[5] < 0 and oth[5] >= 0):
                            continue
                        if xx[j - 1] == _ex(oth, y):
                            off = -1 if y == ymax else 1
                            a, b = _ex(cur, y + off), _ex(oth, y + off)
                            if y == cur[4]:
                                xx[k] = f32(max(a, b) + f32(1)) if cur[5] > 0 else f32(min(a, b) - f32(1))
                            else:
                                xx[k] = f32(min(a, b)) if cur[5] > 0 else f32(max(a, b) + f32(1))
                            break
        xx.sort()
        x_pos = -1 if not xx else 0
        for i in range(1, len(xx), 2):
            x_end = round_down(float(xx[i]))
            if x_end < x_pos:
                continue
            x_start = round_up(float(xx[i - 1]))
            if x_pos > x_start:
                x_start = x_pos
                if x_end < x_start:
                    continue
            _hline(im, x_start, y, x_end)
            x_pos = x_end + 1


def wide_line_edges(x0, y0, x1, y1, width, big, s):
    """Pillow's ImagingDrawWideLine + add_edge: (head, edges) of the quad of one thick segment."""
    dx, dy = x1 - x0, y1 - y0
    small = (width - 1) / 2.0
    rmax, rmin = round_up(small) / big, round_down(small) / big
    dxmin, dxmax = round_down(rmin * dy), round_down(rmax * dy)
    dymin, dymax = round_up(rmin * dx), round_up(rmax * dx)
    v = [(x0 - dxmin, y0 + dymax), (x1 - dxmin, y1 + dymax), (x1 + dxmax, y1 - dymin), (x0 + dxmax, y0 - dymin)]
    edges = []
    for k in range(4):
        (xa, ya), (xb, yb) = v[k], v[(k + 1) % 4]
        dxe = f32(0.0) if ya == yb else f32(xb - xa) / f32(yb - ya)
        edges.append((EDGE, xa, ya, min(ya, yb), max(ya, yb), dxe, min(xa, xb), max(xa, xb)))
    ys = [p[1] for p in v]
    return (QUAD, max(min(ys), 0), min(max(ys), s)), edges


def rasterize(records, flip0, flip1, s, disc_table):
    """records [n,8] int32 -> uint8 keep mask [s,s] (1 = keep, 0 = hole) = rect layer AND NOT flipped brush layer."""
    keep = np.ones((s, s), np.uint8)
    brush = np.zeros((s, s), np.uint8)
    i = 0
    n = len(records)
    while i < n:
        r = records[i]
        t = int(r[0])
        if t == RECT:
            keep[int(r[3]):int(r[4]) + 1, int(r[1]):int(r[2]) + 1] = 0
            i += 1
        elif t == DISC:
            cx, cy, h = int(r[1]), int(r[2]), int(r[3])
            for j in range(2 * h + 1):
                lo, hi = disc_table[h, j]
                if lo <= hi:
                    _hline(brush, cx - h + int(lo), cy - h + j, cx - h + int(hi))
            i += 1
        elif t == POINT:
            if 0 <= r[2] < s and 0 <= r[1] < s:
                brush[int(r[2]), int(r[1])] = 1
            i += 1
        elif t == SEG:
            x0, y0, x1, y1, width = (int(v) for v in r[1:6])
            if x0 == x1 and y0 == y1:
                if 0 <= y0 < s and 0 <= x0 < s:
                    brush[y0, x0] = 1
            else:
                big = float(np.asarray(r[6:8], dtype=np.int32).view(np.float64)[0])
                head, edges = wide_line_edges(x0, y0, x1, y1, width, big, s)
                fill_quad(brush, head, edges)
            i += 1
        elif t == QUAD:
            edges = []
            for e in records[i + 1:i + 5]:
                edges.append((int(e[0]), int(e[1]), int(e[2]), int(e[3]), int(e[4]), np.int32(e[5]).view(np.float32), int(e[6]), int(e[7])))
            fill_quad(brush, r, edges)
            i += 5
        else:
            raise ValueError(f'bad record type {t}')
    if flip0:
        brush = np.flip(brush, 0)
    if flip1:
        brush = np.flip(brush, 1)
    return (keep & (1 - brush)).astype(np.uint8)
