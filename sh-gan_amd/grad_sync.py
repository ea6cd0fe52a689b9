"""Gradient synchronisation for the data-parallel training rows: bucketed all-reduce over RCCL.

The reference trains under ``DistributedDataParallel`` (``lib/experiments/stylegan_default.py:178-187`` wraps every trainable
module, ``broadcast_buffers=False``) and sanitises each gradient before the optimiser step (``nan_to_num(nan=0, posinf=1e5,
neginf=-1e5)``, ``:160-164``).  On MI355X the xGMI fabric is point to point
(7 links x ~153 GB/s per GPU), so a ring all-reduce is bound per link and wants few, large messages -- but one message per phase
cannot overlap with the backward pass.  ``BucketedAllReduce`` therefore
  * lays every parameter's gradient out as a view into a few contiguous fp32 buckets (default 64 MiB: a 58 M-parameter
    discriminator is 4 buckets, the 79 M-parameter generator 5), filled in REVERSE registration order -- the order in which
    backward produces gradients;
  * launches ``all_reduce(bucket, async_op=True)`` from a post-accumulate hook as soon as the last gradient of a bucket has
    been written, i.e. while the rest of backward is still running (RCCL runs on its own stream) -- but ONLY in a backward
    pass that was announced with ``arm()``.  One phase runs several backward passes into the same buckets (Dgen then Dreal,
    Gmain then Gpl, ``effective_batch_gpu`` rounds: ``stylegan_default_loss.py:64,72,93,104,126``) and the reference
    suppresses DDP's reduction in all but the last (``misc.ddp_sync(module, sync)``, ``:32,43,49``); the loss arms the
    buckets before exactly that backward.  Un-armed passes only accumulate;
  * ``finish()`` waits for the outstanding handles, divides by the world size and applies the StyleGAN2 sanitisation
    (``nan_to_num(nan=0, posinf=1e5, neginf=-1e5)``) in one pass per bucket.
One process per GPU, backend ``nccl`` (= RCCL on ROCm); the CPU tests drive the same code over ``gloo``."""
import torch
import torch.distributed as dist


class BucketedAllReduce:
    def __init__(self, params, bucket_bytes=64 << 20, process_group=None, sanitize=True, always_reduce=False):
        # every parameter, whatever its requires_grad flag says NOW: the training stage keeps the networks frozen except inside
        # their own phase (stylegan_default.py:147,157), so the flag is False when the buckets are laid out
        self.params = list(params)
        self.group = process_group
        self.sanitize = sanitize
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.reduce = self.world > 1 or (always_reduce and dist.is_initialized())    # (always_reduce: run the collective in a 1-rank group too)
        self.buckets = []          # flat fp32 tensors
        self._slot = {}            # id(param) -> (bucket index, offset, numel)
        self._pending = []         # per bucket: gradients still missing in this backward pass
        self._handles = []
        self._hooks = []
        self._touched = set()      # id(param) of the parameters that received a gradient since zero_grad()
        self._armed = False        # True only inside the LAST backward pass of a phase (arm())
        self._armed_once = False   # arm() seen since zero_grad()
        self.defer = False         # True: hooks never launch a collective (the backward is being captured / replayed as a HIP graph); reduce_all() does
        cap = max(int(bucket_bytes) // 4, 1)
        cur, cur_n = [], 0
        groups = []
        for p in reversed(self.params):                      # backward order
            if cur and (cur_n + p.numel() > cap or p.device != cur[0].device):
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        for bi, grp in enumerate(groups):
            flat = torch.zeros(sum(p.numel() for p in grp), dtype=torch.float32, device=grp[0].device)
            off = 0
            for p in grp:
                if p.dtype != torch.float32:
                    raise TypeError('BucketedAllReduce: fp32 parameters only')
                p.grad = flat[off:off + p.numel()].view_as(p)          # gradients accumulate straight into the bucket
                self._slot[id(p)] = (bi, off, p.numel())
                off += p.numel()
            self.buckets.append(flat)
            self._pending.append(len(grp))
        self._sizes = list(self._pending)
        for p in self.params:
            was = p.requires_grad                   # (a hook can only be registered while the flag is set; it stays on the tensor)
            p.requires_grad_(True)
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            p.requires_grad_(was)

    # -- hooks ---------------------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        self._touched.add(id(p))
        bi, off, n = self._slot[id(p)]
        flat = self.buckets[bi]
        if p.grad.data_ptr() != flat[off:off + n].data_ptr():            # autograd replaced the view (first accumulation)
            flat[off:off + n].copy_(p.grad.reshape(-1))
            p.grad = flat[off:off + n].view_as(p)
        if not self._armed:                      # an earlier backward of the phase: accumulate only
            return
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and self.reduce and not self.defer:
            self._handles.append((bi, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    # -- per step ------------------------------------------------------------------------------------------------------
    def zero_grad(self):
        """Zero the buckets and (re-)attach every parameter's ``.grad`` view -- an optimiser's ``zero_grad(set_to_none=True)`` drops them."""
        for flat in self.buckets:
            flat.zero_()
        for p in self.params:
            bi, off, n = self._slot[id(p)]
            if p.grad is None or p.grad.data_ptr() != self.buckets[bi][off:off + n].data_ptr():
                p.grad = self.buckets[bi][off:off + n].view_as(p)
        self._pending = list(self._sizes)
        self._handles = []
        self._touched = set()
        self._armed = False
        self._armed_once = False

    def arm(self):
        """Announce the last backward pass of the phase (the reference's ``sync=True`` forward under ``misc.ddp_sync``): from now on
        a bucket is reduced as soon as every parameter in it has received THIS pass's gradient.  Buckets holding a parameter the
        pass does not reach are reduced in ``finish()``."""
        if self._armed:
            raise RuntimeError('BucketedAllReduce.arm(): already armed -- finish() must run between two armed backward passes')
        self._armed = True
        self._armed_once = True
        self._pending = list(self._sizes)

    def was_armed(self):
        """True if ``arm()`` was called since the last ``zero_grad()`` / ``finish()``."""
        return self._armed or self._armed_once

    def untouched(self):
        """Parameters that received no gradient since ``zero_grad()`` -- their bucket slots hold zeros; an optimiser that must skip
        them (``zero_grad(set_to_none=True)`` semantics: no moment decay, no step) gets ``p.grad = None`` for these."""
        return [p for p in self.params if id(p) not in self._touched]

    def reduce_all(self):
        """Every bucket now, between two graph replays (train_stage.PhaseGraphs with more than one rank: the backward passes of a phase are
        one HIP graph, no hook runs on the host while it replays, so nothing can be launched bucket by bucket under it).  All buckets are
        in flight together (RCCL: its own stream; the calling stream waits for them); follow with ``finish(reduced=True)``."""
        if not self.reduce:
            return
        hs = [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for flat in self.buckets]
        for h in hs:
            h.wait()

    def finish(self, reduced=False):
        """Wait for the reductions launched by the armed backward pass; average, sanitise.  Buckets that were not launched (a
        parameter the armed pass did not reach, or no armed pass at all) are reduced here, synchronously.  ``reduced=True``: the caller
        has run ``reduce_all()`` -- only the averaging and the sanitisation are left (this part can be captured in a graph)."""
        launched = {bi for bi, _ in self._handles}
        for bi, h in self._handles:
            h.wait()
        for bi, flat in enumerate(self.buckets):
            if self.reduce and not reduced and bi not in launched:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.world > 1:
                flat.div_(self.world)
            if self.sanitize:
                torch.nan_to_num(flat, nan=0.0, posinf=1e5, neginf=-1e5, out=flat)
        self._handles = []
        self._pending = list(self._sizes)
        self._armed = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
