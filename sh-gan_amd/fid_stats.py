"""FID statistics around the generator (reference lib/evaluator/eva_fid.py:194-277 and the
broadcast-based ``sync`` of lib/evaluator/eva_base.py:96-188).

What the reference does per batch: every rank broadcasts its [B,2048] features (and file names) to every other rank with
3 x world ``dist.broadcast`` calls, rank 0 keeps python lists of float64 arrays and at the end computes
``mu, sigma = mean, x^T x / n - mu mu^T`` in numpy and ``sqrtm`` in scipy.

MI355X-first form:
  * ``FidStats``: float64 second moments of the augmented feature [x, 1] accumulated ON the device by an fp64-MFMA kernel
    (csrc/fid_stats.hip); samples that exist only because ``DistributedSampler(extend=True)`` padded the last round are
    given weight 0 (the reference drops them with ``[0:sample_n]`` after re-interleaving);
  * ONE ``all_reduce`` of the [D+1, D+1] moments (33.6 MB of float64 at D = 2048, RCCL over xGMI) at the end of the
    evaluation replaces all per-batch collectives;
  * ``gather_features``: the reference's own semantics when the features themselves are wanted in dataset order --
    one ``all_gather_into_tensor`` + the zipzap re-interleave instead of 3 x world broadcasts;
  * ``fid_from_stats``: the host tail (``scipy.linalg.sqrtm``), as eva_fid.py:259-261.
The Inception-v3 feature detector itself is a TorchScript download (eva_fid.py:30,145-158) and cannot be pinned offline:
everything from the [B,2048] features onwards is implemented and tested here."""
import numpy as np
import torch

from . import _lib, kernels
from ._lib import check
from .data import zipzap_arrange


class FidStats:
    def __init__(self, dim=2048, device='cuda', accumulate_fn=None):
        self.dim = int(dim)
        self.dp = (self.dim + 1 + 31) // 32 * 32
        self.device = torch.device(device)
        self.S = torch.zeros((self.dp, self.dp), dtype=torch.float64, device=self.device)
        self._accumulate = accumulate_fn            # tests inject a numpy stand-in on CPU; the product path is the HIP kernel

    def add(self, feats, weights=None):
        """feats [B, dim] float32 / float64 on the device; weights [B] float32 (1 = count the sample, 0 = padded duplicate)."""
        if self._accumulate is not None:
            self._accumulate(self.S, feats, weights)
            return
        L = kernels._Launch()
        feats = L.req(feats, 'feats', dtype=feats.dtype if isinstance(feats, torch.Tensor) and feats.dtype == torch.float64 else torch.float32)
        weights = L.req(weights, 'weights')
        L.req(self.S, 'S', dtype=torch.float64)
        if feats.ndim != 2 or feats.shape[1] != self.dim:
            raise _lib.ShgError(f'FidStats.add: features must be [B, {self.dim}]')
        with L:
            check(_lib.get_lib().shg_fid_accumulate_f64(kernels._ptr(feats), int(feats.dtype == torch.float64), kernels._ptr(weights),
                                                        kernels._ptr(self.S), feats.shape[0], self.dim, self.dp, L.stream()), 'fid_accumulate')

    def add_shard(self, feats, k0, rank, world, sample_n):
        """Features of this rank's items k0 .. k0+B-1 (positions in its ``DistributedSampler(extend=True)`` list): item k of
        rank r sits at position k*world + r of the re-interleaved list, which the reference truncates to ``sample_n``."""
        b = feats.shape[0]
        pos = (torch.arange(k0, k0 + b, device=feats.device) * world + rank)
        self.add(feats, (pos < sample_n).to(torch.float32))

    def add_images(self, detector, images, k0=None, rank=0, world=1, sample_n=None):
        """The detector hand-off of ``fid_evaluator.add_batch`` (eva_fid.py:194-206): ``images`` [B,3,H,W] in 0..255 (the uint8
        composite of ``run_generator`` or ``real*127.5+127.5``, shgan_default.py:281-288) -> ``detector(images.float(),
        return_features=True)`` -> moments.  ``detector`` is the caller's Inception-v3 TorchScript module (a download,
        eva_fid.py:20,145-158: not shipped and not reproduced); any callable with that signature works.  With ``k0`` the batch
        is this rank's items k0.. of its sampler list and padded duplicates beyond ``sample_n`` get weight 0 (``add_shard``)."""
        with torch.no_grad():
            feats = detector(images.to(self.device).float(), return_features=True)
        if feats.ndim != 2 or feats.shape[1] != self.dim:
            raise _lib.ShgError(f'FidStats.add_images: the detector returned {tuple(feats.shape)}, expected [B, {self.dim}]')
        if k0 is None:
            self.add(feats)
        else:
            self.add_shard(feats, k0, rank, world, sample_n)
        return feats

    def all_reduce(self):
        """Sum the moments over all ranks (one collective per evaluation)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():     # (a 1-rank group goes through the collective too: same path as N ranks)
            if dist.get_backend() == 'gloo' and self.S.is_cuda:   # ranks sharing one device under gloo: the collective takes host tensors
                h = self.S.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                self.S.copy_(h)
            else:
                dist.all_reduce(self.S, op=dist.ReduceOp.SUM)
        return self

    def mean_cov(self, sample_n=None):
        """-> (n, mu [dim], sigma [dim, dim]) numpy float64; ``mu = mean``, ``sigma = x^T x / sample_n - mu mu^T`` as
        eva_fid.py:252-255 (the reference divides the second moment by ``sample_n``, not by the row count; they differ only
        when fewer than ``sample_n`` features exist -- default: the accumulated count)."""
        S = self.S.cpu().numpy()
        S = np.triu(S) + np.triu(S, 1).T                     # the kernel keeps tiles on / above the diagonal
        d = self.dim
        n = float(S[d, d])
        if not n > 0:
            raise ValueError('FidStats.mean_cov: no samples accumulated (empty accumulator, or every sample had weight 0)')
        mu = S[:d, d] / n
        sigma = S[:d, :d] / (n if sample_n is None else float(sample_n)) - np.outer(mu, mu)
        return n, mu, sigma


def fid_from_stats(mu_fake, sigma_fake, mu_real, sigma_real):
    """eva_fid.py:258-261."""
    import scipy.linalg
    m = np.square(mu_fake - mu_real).sum()
    s, _ = scipy.linalg.sqrtm(np.dot(sigma_fake, sigma_real), disp=False)
    return float(np.real(m + np.trace(sigma_fake + sigma_real - s * 2)))


def gather_features(local_feats, n_items, rank, world):
    """All ranks' [B_r, D] features -> numpy [n_items, D] in dataset order: ``all_gather_into_tensor`` + zipzap
    (replaces ``base_evaluator.sync`` + ``zipzap_arrange``, eva_fid.py:217-229).  Every rank must hold the same count
    (DistributedSampler(extend=True) guarantees it)."""
    import torch.distributed as dist
    if world == 1:
        return local_feats.detach().cpu().numpy()[:n_items]
    full = torch.empty((world * local_feats.shape[0],) + tuple(local_feats.shape[1:]), dtype=local_feats.dtype, device=local_feats.device)
    dist.all_gather_into_tensor(full, local_feats.contiguous())
    full = full.view((world,) + tuple(local_feats.shape))
    return zipzap_arrange([full[r].cpu().numpy() for r in range(world)])[:n_items]
