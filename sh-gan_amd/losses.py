"""The StyleGAN2 training loss on the HIP training path -- same class name, constructor and ``accumulate_gradients`` contract as
``lib/experiments/stylegan_default_loss.py:16-128`` (non-saturating logistic loss, lazy R1 on the discriminator, lazy
path-length regularisation on the generator, style mixing), so that the reference's training loop
(``stylegan_default.py:150-166``: one call per phase and round, then ``nan_to_num`` + optimiser step) can drive it unchanged.

What differs: gradients flow through this package's differentiable operators (``model_zoo/stylegan_utils/conv2d_gradfix.py``,
``grad_ops.py``, ``upfirdn2d.py`` -- convolution, FIR and activation forward and backward in HIP, twice differentiable);
``sync`` keeps the reference's meaning -- only the LAST backward pass of a phase call with ``sync=True`` may reduce gradients
across ranks (``misc.ddp_sync``: Gmain ``sync and not do_Gpl``, Gpl ``sync``, Dgen never, Dreal ``sync``; ``:56-106``) -- and is
delivered to ``grad_sync.BucketedAllReduce`` as ``self.grad_sync.arm()`` right before that backward (``grad_sync`` is set by the
training stage per phase; None = single process); an ADA ``augment_pipe`` is not supported; the statistics reporting of the
reference (``training_stats.report``) is replaced by ``self.stats``, a plain dict of the last values."""
import numpy as np
import torch
import torch.nn.functional as F

from .model_zoo.stylegan_utils import conv2d_gradfix

PHASES = ('Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth')


class Loss:
    """Contract with the training stage (train_stage.run_phases): one call per phase and round.  ``sync=True`` marks the phase's last
    round; the implementation must call ``self.grad_sync.arm()`` (when ``grad_sync`` is not None) right before the LAST ``backward()`` of
    that call and before no other -- the buckets then all-reduce under that backward pass.  Without it the gradients are still reduced
    (synchronously, in ``finish()``) and the stage says so once."""
    grad_sync = None

    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync, gain):
        raise NotImplementedError()


class StyleGAN2Loss(Loss):
    def __init__(self, device, G_mapping, G_synthesis, D, augment_pipe=None, style_mixing_prob=0.9, r1_gamma=10, pl_batch_shrink=2,
                 pl_decay=0.01, pl_weight=2, batch_critic=True):
        super().__init__()
        self.batch_critic = batch_critic            # Dmain: generated and real images through the critic as ONE batch (see accumulate_gradients)
        if augment_pipe is not None:
            raise NotImplementedError('ADA augmentation is not part of the HIP path')
        self.device = device
        self.G_mapping, self.G_synthesis, self.D = G_mapping, G_synthesis, D
        self.style_mixing_prob = style_mixing_prob
        self.r1_gamma, self.pl_batch_shrink, self.pl_decay, self.pl_weight = r1_gamma, pl_batch_shrink, pl_decay, pl_weight
        self.pl_mean = torch.zeros([], device=device)
        self.randn_like = torch.randn_like          # (tests substitute a fixed draw for the path-length noise)
        self.stats = {}
        self.grad_sync = None                       # the running phase's BucketedAllReduce (train_stage.run_phases sets it)

    def _arm(self, sync):
        if sync and self.grad_sync is not None:
            self.grad_sync.arm()

    # -- forward helpers (stylegan_default_loss.py:31-51) ----------------------------------------------------------------
    def _style_mix(self, ws, z, c):
        if self.style_mixing_prob > 0:
            # with probability p the rows from a random cutoff on come from a second latent
            cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, ws.shape[1])
            cutoff = torch.where(torch.rand([], device=ws.device) < self.style_mixing_prob, cutoff, torch.full_like(cutoff, ws.shape[1]))
            # the second mapping always runs (stylegan_default_loss.py:39: same consumption of the device RNG stream whether or
            # not mixing is drawn) and the rows are selected on the device -- no host sync on the cutoff
            ws2 = self.G_mapping(torch.randn_like(z), c, skip_w_avg_update=True)
            rows = torch.arange(ws.shape[1], device=ws.device).reshape(1, -1, 1)
            ws = torch.where(rows >= cutoff, ws2, ws)
        return ws

    def run_G(self, z, c, sync=True):
        ws = self._style_mix(self.G_mapping(z, c), z, c)
        return self.G_synthesis(ws), ws

    def run_D(self, img, c, sync=True):
        return self.D(img, c)

    def run_D_pair(self, gen_img, gen_c, real_img, real_c):
        """-> (gen_logits, real_logits) from one critic pass over cat([gen, real]) with the minibatch statistic per half."""
        logits = self.D(torch.cat([gen_img, real_img]), torch.cat([gen_c, real_c]), segments=2)
        return logits[:gen_img.shape[0]], logits[gen_img.shape[0]:]

    # -- one phase (stylegan_default_loss.py:53-128) -----------------------------------------------------------------------
    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync=True, gain=1):
        if phase not in PHASES:
            raise AssertionError(f'unknown phase {phase!r}')
        do_Gmain = phase in ('Gmain', 'Gboth')
        do_Dmain = phase in ('Dmain', 'Dboth')
        do_Gpl = phase in ('Greg', 'Gboth') and self.pl_weight != 0
        do_Dr1 = phase in ('Dreg', 'Dboth') and self.r1_gamma != 0
        with torch.enable_grad():
            if do_Gmain:                                             # maximise the logits of generated images
                gen_img, _ = self.run_G(gen_z, gen_c)
                gen_logits = self.run_D(gen_img, gen_c)
                loss_Gmain = F.softplus(-gen_logits)                 # -log(sigmoid(logits))
                self.stats.update({'Loss/scores/fake': gen_logits.detach(), 'Loss/G/loss': loss_Gmain.detach()})
                self._arm(sync and not do_Gpl)                       # (may get synced by Gpl, :56)
                loss_Gmain.mean().mul(gain).backward()

            if do_Gpl:                                               # path-length regularisation on a shrunk batch
                n = gen_z.shape[0] // self.pl_batch_shrink
                gen_img, gen_ws = self.run_G(gen_z[:n], gen_c[:n])
                pl_noise = self.randn_like(gen_img) / np.sqrt(gen_img.shape[2] * gen_img.shape[3])
                with conv2d_gradfix.no_weight_gradients():
                    (pl_grads,) = torch.autograd.grad(outputs=[(gen_img * pl_noise).sum()], inputs=[gen_ws], create_graph=True,
                                                      only_inputs=True)
                pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
                pl_mean = self.pl_mean.lerp(pl_lengths.mean(), self.pl_decay)
                self.pl_mean.copy_(pl_mean.detach())
                pl_penalty = (pl_lengths - pl_mean).square()
                loss_Gpl = pl_penalty * self.pl_weight
                self.stats.update({'Loss/pl_penalty': pl_penalty.detach(), 'Loss/G/reg': loss_Gpl.detach()})
                self._arm(sync)
                (gen_img[:, 0, 0, 0] * 0 + loss_Gpl).mean().mul(gain).backward()

            if do_Dmain and not do_Dr1 and self.batch_critic and real_img.shape[0] == gen_z.shape[0]:
                # The reference judges the generated and the real batch in two critic passes with a backward each (:84-106).  No layer of
                # the critic mixes samples except the minibatch statistic, so one pass over the stacked batch (statistic per half) and one
                # backward of the summed losses give the same logits and the same gradients (to the order of the weight-gradient sums)
                # with half the weight-side work and half the launches.  Dreg keeps the reference's form (R1 needs the real pass alone).
                with torch.no_grad():
                    gen_img, _ = self.run_G(gen_z, gen_c)
                gen_logits, real_logits = self.run_D_pair(gen_img, gen_c, real_img.detach(), real_c)
                loss_Dgen, loss_Dreal = F.softplus(gen_logits), F.softplus(-real_logits)
                self.stats.update({'Loss/scores/fake': gen_logits.detach(), 'Loss/scores/real': real_logits.detach(),
                                   'Loss/D/loss': (loss_Dgen + loss_Dreal).detach()})
                self._arm(sync)
                (loss_Dgen.mean() + loss_Dreal.mean()).mul(gain).backward()
                return

            loss_Dgen = 0
            if do_Dmain:                                             # minimise the logits of generated images
                with torch.no_grad():
                    gen_img, _ = self.run_G(gen_z, gen_c)
                gen_logits = self.run_D(gen_img, gen_c)
                loss_Dgen = F.softplus(gen_logits)                   # -log(1 - sigmoid(logits))
                self.stats['Loss/scores/fake'] = gen_logits.detach()
                loss_Dgen.mean().mul(gain).backward()

            if do_Dmain or do_Dr1:                                   # maximise the logits of real images; R1 on the same pass
                real_tmp = real_img.detach().requires_grad_(do_Dr1)
                real_logits = self.run_D(real_tmp, real_c)
                self.stats['Loss/scores/real'] = real_logits.detach()
                loss_Dreal = 0
                if do_Dmain:
                    loss_Dreal = F.softplus(-real_logits)
                    self.stats['Loss/D/loss'] = (loss_Dgen + loss_Dreal).detach()
                loss_Dr1 = 0
                if do_Dr1:
                    with conv2d_gradfix.no_weight_gradients():
                        (r1_grads,) = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_tmp], create_graph=True,
                                                          only_inputs=True)
                    r1_penalty = r1_grads.square().sum([1, 2, 3])
                    loss_Dr1 = (r1_penalty * (self.r1_gamma / 2)).reshape(-1, 1)
                    self.stats.update({'Loss/r1_penalty': r1_penalty.detach(), 'Loss/D/reg': loss_Dr1.detach()})
                self._arm(sync)                                      # Dgen's backward above never syncs (:87)
                (real_logits * 0 + loss_Dreal + loss_Dr1).mean().mul(gain).backward()


class InpaintingLoss(StyleGAN2Loss):
    """The same four phases for the co-modulated inpainting generator of SH-GAN (BASELINE config 5: FFHQ-512 G + D step).  The
    reference ships no training stage for ``shgan``, so this is formula-level: the phase algebra is
    ``stylegan_default_loss.py:53-128`` unchanged; what differs is how the two networks are fed, following CoModGAN:
      * ``real_img`` is the discriminator's real input ``[N, 4, R, R] = cat([mask - 0.5, real])`` (``ic_n = 4``);
      * the generator input is derived from it as the evaluation loop does (``shgan_default.py:268-273``):
        ``x = cat([mask - 0.5, real * mask])``, ``img = G.synthesis(*G.encoder(x), ws)`` with ``ws`` from ``G.mapping`` (style mixing on ``ws``);
      * generated images reach the discriminator as ``cat([mask - 0.5, img])``; R1 differentiates the logits w.r.t. the 4-channel real input;
      * ``composite_fake`` (REQUIRED keyword, no default: the two values are two training objectives and neither may be picked
        silently; False = the reference's phase algebra on the raw generator output, the mode the reference-autograd fixtures pin;
        True = the objective to train an inpainting generator with): when True the generator's output is composited with the known pixels,
        ``img * (1 - mask) + real * mask``, before anything downstream sees it -- the critic, and with it the path-length
        regulariser -- as CoModGAN does inside its generator and as the reference's evaluation loop does with this generator
        (``shgan_default.py:259``).  Without it the critic can tell real from fake from the known region alone and the generator is
        trained on pixels that are thrown away at evaluation time.  ``composite_fake=False`` is the raw-output objective; the
        full-width reference-autograd fixtures (``tests/golden/config5_step512.npz``, generated by feeding the reference's modules
        the raw output) are evaluated in that mode, the composite itself is pinned by ``tests/test_gpu_config5.py``.
    CoModGAN's optional L1 term on the known region is not part of ``stylegan_default_loss.py`` and is not added.
    ``noise_mode`` is the synthesis noise ('random' in training; tests pin 'const')."""

    def __init__(self, device, G, D, *, composite_fake, noise_mode='random', **kw):
        super().__init__(device, G.mapping, None, D, **kw)
        self.G = G
        self.noise_mode = noise_mode
        self.composite_fake = composite_fake
        self._m05 = self._x = None

    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync=True, gain=1):
        if real_img.ndim != 4 or real_img.shape[1] != 4:
            raise AssertionError('InpaintingLoss: real_img must be [N, 4, R, R] = cat([mask - 0.5, real])')
        self._m05 = real_img[:, 0:1].detach()
        self._x = torch.cat([self._m05, real_img[:, 1:4].detach() * (self._m05 + 0.5)], dim=1)
        try:
            super().accumulate_gradients(phase, real_img, real_c, gen_z, gen_c, sync=sync, gain=gain)
        finally:
            self._m05 = self._x = None

    def run_G(self, z, c, sync=True):
        n = z.shape[0]
        ws = self._style_mix(self.G.mapping(z, c), z, c)
        x_global, feats = self.G.encoder(self._x[:n])
        img = self.G.synthesis(x_global, feats, ws, noise_mode=self.noise_mode)
        if self.composite_fake:                                  # known pixels from the input (x[:, 1:4] = real * mask), the hole from G
            m = self._m05[:n] + 0.5
            img = img * (1.0 - m) + self._x[:n, 1:4]
        return img, ws

    def run_D(self, img, c, sync=True):
        if img.shape[1] == 3:                                   # a generated image: prepend the mask channel it was conditioned on
            img = torch.cat([self._m05[:img.shape[0]], img], dim=1)
        return self.D(img, c)

    def run_D_pair(self, gen_img, gen_c, real_img, real_c):
        n = gen_img.shape[0]
        both = torch.cat([torch.cat([self._m05[:n], gen_img], dim=1), real_img])       # (one write pass: cat of the cat's pieces)
        logits = self.D(both, torch.cat([gen_c, real_c]), segments=2)
        return logits[:n], logits[n:]
