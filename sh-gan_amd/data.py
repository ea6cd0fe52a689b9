"""Integer paths that feed / order the generator in the eval loop:
freeform mask synthesis (lib/data_factory/ds_ffhq.py:145-217), rank-strided sample sharding
(lib/data_factory/common/ds_sampler.py:43-68) and the re-interleave of rank-sharded results
(lib/evaluator/eva_base.py:196-230).  Host-side numpy / PIL, bit-exact with the reference for a
given numpy global-RNG state (the stroke rasteriser is Pillow's, as in the reference)."""
import math

import numpy as np
import torch


def RandomBrush(max_tries, s, min_num_vertex=4, max_num_vertex=18, mean_angle=2 * math.pi / 5,
                angle_range=2 * math.pi / 15, min_width=12, max_width=48):
    """Random thick poly-line strokes on an s x s canvas -> uint8 {0,1} array."""
    from PIL import Image, ImageDraw
    mean_radius = math.sqrt(s * s + s * s) / 8
    canvas = Image.new('L', (s, s), 0)
    rng = np.random
    for _ in range(rng.randint(max_tries)):
        n_vertex = rng.randint(min_num_vertex, max_num_vertex)
        lo = mean_angle - rng.uniform(0, angle_range)
        hi = mean_angle + rng.uniform(0, angle_range)
        turn = [(2 * math.pi - rng.uniform(lo, hi)) if (i % 2 == 0) else rng.uniform(lo, hi) for i in range(n_vertex)]
        cw, ch = canvas.size
        path = [(int(rng.randint(0, cw)), int(rng.randint(0, ch)))]
        for ang in turn:
            step = np.clip(rng.normal(loc=mean_radius, scale=mean_radius // 2), 0, 2 * mean_radius)
            px = np.clip(path[-1][0] + step * math.cos(ang), 0, cw)
            py = np.clip(path[-1][1] + step * math.sin(ang), 0, ch)
            path.append((int(px), int(py)))
        pen = ImageDraw.Draw(canvas)
        thick = int(rng.uniform(min_width, max_width))
        pen.line(path, fill=1, width=thick)
        half = thick // 2
        for (vx, vy) in path:
            pen.ellipse((vx - half, vy - half, vx + half, vy + half), fill=1)
        # the reference draws two flip decisions here and discards the flipped copies (ds_ffhq.py:188-191)
        rng.random()
        rng.random()
    arr = np.asarray(canvas, np.uint8)
    if rng.random() > 0.5:
        arr = np.flip(arr, 0)
    if rng.random() > 0.5:
        arr = np.flip(arr, 1)
    return arr


def RandomMask(s, hole_range=[0, 1]):
    """1 = keep, 0 = hole.  Rectangular holes AND-ed with the complement of brush strokes; resampled
    until the hole ratio lies strictly inside ``hole_range``.  -> float32 [1,s,s]."""
    coef = min(hole_range[0] + hole_range[1], 1.0)
    rng = np.random
    while True:
        keep = np.ones((s, s), np.uint8)

        def punch(max_size):
            w, h = rng.randint(max_size), rng.randint(max_size)
            x, y = rng.randint(-(w // 2), s - w + w // 2), rng.randint(-(h // 2), s - h + h // 2)
            keep[max(y, 0): min(y + h, s), max(x, 0): min(x + w, s)] = 0

        for max_tries, max_size in ((int(10 * coef), s // 2), (int(5 * coef), s)):
            for _ in range(rng.randint(max_tries)):
                punch(max_size)
        keep = np.logical_and(keep, 1 - RandomBrush(int(20 * coef), s))
        hole = 1 - np.mean(keep)
        if hole_range is not None and (hole <= hole_range[0] or hole >= hole_range[1]):
            continue
        return keep[np.newaxis, ...].astype(np.float32)


class DistributedSampler(torch.utils.data.Sampler):
    """Rank-strided index shard: indices[rank::world] of the (optionally shuffled) index list, padded
    with its leading entries (``extend``) or truncated so every rank gets the same count."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True, extend=False):
        import torch.distributed as dist
        if num_replicas is None:
            num_replicas = dist.get_world_size()
        if rank is None:
            rank = dist.get_rank()
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        per_rank = len(dataset) // num_replicas
        if extend and len(dataset) != per_rank * num_replicas:
            per_rank += 1
        self.num_samples = per_rank
        self.total_size = per_rank * num_replicas
        self.shuffle, self.extend = shuffle, extend

    def get_sync_order(self):
        if not self.shuffle:
            return list(range(len(self.dataset)))
        import torch.distributed as dist
        order = torch.randperm(len(self.dataset))
        if dist.is_available() and dist.is_initialized():
            dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
            order = order.to(dev)
            dist.broadcast(order, src=0)       # every rank adopts rank 0's permutation
        return order.cpu().tolist()

    def __iter__(self):
        order = self.get_sync_order()
        order = (order + order[0: self.total_size - len(order)]) if self.extend else order[0: self.total_size]
        return iter(order[self.rank: len(order): self.num_replicas])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        pass


def zipzap_arrange(data):
    """[[0,2,4,6],[1,3,5,7]] (one list / array per rank) -> dataset order [0,1,2,...]."""
    if isinstance(data[0], list):
        total = sum(len(d) for d in data)
        out = []
        for i in range(max(len(d) for d in data)):
            for d in data:
                if i < len(d) and len(out) < total:
                    out.append(d[i])
        return out
    if isinstance(data[0], np.ndarray):
        total = sum(d.shape[0] for d in data)
        longest = max(d.shape[0] for d in data)
        tail = data[0].shape[1:]
        padded = [np.concatenate([d, np.zeros((longest - d.shape[0],) + tail, d.dtype)], axis=0) if d.shape[0] < longest else d
                  for d in data]
        return np.stack(padded, axis=1).reshape((-1,) + tail)[:total]
    if data[0] is None:
        return list(data)
    raise NotImplementedError
