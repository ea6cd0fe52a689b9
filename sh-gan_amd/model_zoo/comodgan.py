"""Co-modulated GAN generator (encoder + co-modulated synthesis), host side.  Mirrors the registry
names, constructor arguments, forward signatures and state_dict schema of the reference's
lib/model_zoo/comodgan.py (``Mapping`` :30, ``encoder_block`` :34, ``encoder_epilogue`` :66,
``Encoder`` :115, ``synthesis_block_first`` :207, ``synthesis_block`` :264, ``Synthesis`` :342,
``Generator`` :435); every tensor op runs on libshgan_hip kernels."""
import numpy as np
import torch
import torch.nn as nn

from .. import kernels
from .common.get_model import get_model, register
from .stylegan import Generator as Generator_StyleGan
from .stylegan import Mapping as Mapping_StyleGan
from .stylegan import _add, dense, discrim_block, discrim_epilogue, synthesis_layer, torgb_layer
from .stylegan import synthesis_block as stylegan_synthesis_block
from . import stylegan as stylegan_mod
from .stylegan_utils import grad_ops, upfirdn2d

version = '0'
symbol = 'comodgan'

_ACT = 'lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)'


@register('comodgan_mapping')
class Mapping(Mapping_StyleGan):
    pass


class encoder_block(discrim_block):
    """Down block that also returns its pre-downsample feature map (comodgan.py:38-64)."""

    def forward(self, x, img):
        x = grad_ops.to_block_dtype(x, self.use_fp16)                            # comodgan.py:39-43
        if self.fromrgb is not None:
            y = self.fromrgb(grad_ops.to_block_dtype(img, self.use_fp16))        # :45-50
            x = _add(x, y) if x is not None else y
        if self.reslink:
            y = self.skip(x, gain=np.sqrt(0.5))
            feat = self.conv0(x)
            x = _add(self.conv1(feat, gain=np.sqrt(0.5)), y)
        else:
            feat = self.conv0(x)
            if stylegan_mod.JOIN_INPUT_GRADS and grad_ops.wants_grad(feat) and feat.is_cuda:
                # training rows: the feature map feeds conv1 and, later, the synthesis network; the gradient that comes back from there
                # first is added by conv1's input-gradient kernel in its store pass (grad_ops.InputGradJoin)
                join = grad_ops.InputGradJoin(feat)
                with grad_ops.InputGradJoin.consumer(join):
                    x = self.conv1(feat)
                return x, None, grad_ops.stash_input_grad(feat, join)
            x = self.conv1(feat)
        return x, None, feat


class encoder_epilogue(discrim_epilogue):
    """4x4 tail of the encoder: conv -> flatten -> fc -> [out] -> dropout (comodgan.py:66-113)."""

    def __init__(self, ic_n, oc_n, resolution, cmap_dim, rgb_n=None, mbstd_group_size=4, mbstd_c_n=1, activation=_ACT,
                 reslink=True, use_dropout=True, has_extra_final_layer=True):
        super().__init__(ic_n=ic_n, resolution=resolution, cmap_dim=cmap_dim, rgb_n=rgb_n, mbstd_group_size=mbstd_group_size,
                         mbstd_c_n=mbstd_c_n, activation=activation, reslink=reslink)
        self.fc = dense(ic_n * (resolution ** 2), oc_n, activation=activation)
        self.out = dense(oc_n, oc_n, activation=None) if has_extra_final_layer else None
        self.dropout = nn.Dropout(p=0.5) if use_dropout else None

    def forward(self, x, img=None, cmap=None):
        x = grad_ops.to_block_dtype(x, False)                                    # comodgan.py:99: the tail is always float32
        if self.fromrgb is not None:
            x = _add(x, self.fromrgb(img.to(torch.float32)))
        if self.mbstd is not None:
            x = self.mbstd(x)
        feat = self.conv(x)
        x = self.fc(feat.flatten(1))
        if self.out is not None:
            x = self.out(x)
        if self.dropout is not None and self.training:
            x = self.dropout(x)                                  # comodgan.py:109-110 (identity in eval(); [N, oc_n] -- a tensor op)
        if self.cmap_dim is not None:
            raise NotImplementedError('conditional projection is not on the generator path')
        return x, feat


@register('comodgan_encoder', version)
class Encoder(nn.Module):
    def __init__(self, resolution=256, ic_n=3, oc_n=1024, ch_base=16384, ch_max=512, use_fp16_before_res=16,
                 resample_filter=[1, 3, 3, 1], activation=_ACT, mbstd_group_size=4, mbstd_c_n=1, c_dim=None, cmap_dim=None,
                 use_dropout=True, has_extra_final_layer=True):
        super().__init__()
        log2res = int(np.log2(resolution))
        if 2 ** log2res != resolution:
            raise ValueError
        if c_dim is not None and c_dim > 0:
            raise NotImplementedError('label-conditioned encoders are not on the SH-GAN path')
        self.encode_res = [2 ** i for i in range(log2res, 1, -1)]
        self.ic_n, self.ch_base, self.ch_max = ic_n, ch_base, ch_max
        self.resample_filter, self.activation = resample_filter, activation
        for idx, (ri, rj) in enumerate(zip(self.encode_res[:-1], self.encode_res[1:])):
            ci, cj = min(ch_base // ri, ch_max), min(ch_base // rj, ch_max)
            setattr(self, 'b{}'.format(ri), encoder_block(ci, ci, cj, rgb_n=(ic_n if idx == 0 else None),
                                                        resample_filter=resample_filter, activation=activation, reslink=False,
                                                        use_fp16=(use_fp16_before_res is not None and ri > use_fp16_before_res)))   # comodgan.py:149
        self.mapping = None
        c4 = min(ch_base // self.encode_res[-1], ch_max)
        self.b4 = encoder_epilogue(c4, oc_n, resolution=4, cmap_dim=None, activation=activation,
                                   mbstd_group_size=mbstd_group_size, mbstd_c_n=mbstd_c_n, reslink=False,
                                   use_dropout=use_dropout, has_extra_final_layer=has_extra_final_layer)

    def forward(self, img, c=None):
        x = None
        feats = {}
        for res in self.encode_res[0:-1]:
            x, img, feat = getattr(self, 'b{}'.format(res))(x, img)
            feats[res] = feat
        x, feat = self.b4(x, img, None)
        feats[4] = feat
        return x, feats


def _style_of(cache, layer):
    return None if cache is None else cache.get(id(layer))


def _style_in(cache, layer, w_iter, w0):
    """The layer's style input cat([w_i, x_global]) (comodgan.py:253,258) -- skipped when its styles come from the cache."""
    w = next(w_iter)
    if cache is not None and id(layer) in cache:
        return None
    return torch.cat([w, w0], dim=1)


class synthesis_block_first(nn.Module):
    """4x4 block: fc(x_global) reshaped + encoder feature, one modulated conv, torgb (comodgan.py:207-262)."""

    def __init__(self, w0_dim, oc_n, w_dim, resolution, rgb_n=None, activation=_ACT):
        super().__init__()
        self.resolution = resolution
        self.fc = dense(w0_dim, oc_n * (resolution ** 2), activation=activation)
        self.num_conv = 0
        self.num_torgb = 0
        self.conv = synthesis_layer(oc_n, oc_n, 3, w0_dim + w_dim, resolution=4, bias=True, activation=activation)
        self.num_conv += 1
        self.torgb = None
        if rgb_n is not None:
            self.torgb = torgb_layer(oc_n, rgb_n, 1, w0_dim + w_dim, activation=None)
            self.num_torgb += 1

    def forward(self, x, x0, ws, fused_modconv=None, noise_mode='random', style_cache=None):
        w0 = x.to(torch.float32)
        x = self.fc(w0).view(w0.size(0), -1, self.resolution, self.resolution)
        x = _add(x, x0)
        w_iter = iter(ws.unbind(dim=1))
        x = self.conv(x, _style_in(style_cache, self.conv, w_iter, w0), noise_mode=noise_mode, styles_sd=_style_of(style_cache, self.conv))
        img = None
        if self.torgb is not None:
            img = self.torgb(x, _style_in(style_cache, self.torgb, w_iter, w0), styles_sd=_style_of(style_cache, self.torgb))
        return x, img


class synthesis_block(stylegan_synthesis_block):
    """up-conv (+ encoder skip) -> conv -> RGB skip, every style = affine(cat[w, x_global]) (comodgan.py:264-340)."""

    def __init__(self, ic_n, oc_n, w_dim, w0_dim, resolution, rgb_n, resample_filter=[1, 3, 3, 1], activation=_ACT,
                 res_link=False, use_fp16=False):
        if ic_n == 0:
            raise ValueError
        super().__init__(ic_n, oc_n, w_dim, resolution, rgb_n, resample_filter, activation, res_link, use_fp16)
        self.conv0 = synthesis_layer(ic_n, oc_n, 3, w_dim=w_dim + w0_dim, resolution=resolution, up=2, activation=activation,
                                     resample_filter=resample_filter, use_noise=True)
        self.conv1 = synthesis_layer(oc_n, oc_n, 3, w_dim=w_dim + w0_dim, resolution=resolution, up=1, activation=activation,
                                     resample_filter=None, use_noise=True)
        if self.torgb is not None:
            self.torgb = torgb_layer(oc_n, rgb_n, 1, w_dim=w_dim + w0_dim, activation=None)

    def forward(self, x, x0, img, ws, w0, fused_modconv=None, noise_mode='random', style_cache=None):
        x, x0 = grad_ops.to_block_dtype(x, self.use_fp16), grad_ops.to_block_dtype(x0, self.use_fp16)      # comodgan.py:305-312
        fm = stylegan_mod.fused_modconv_rule(self, x, fused_modconv)                                         # comodgan.py:307-309
        w_iter = iter(ws.unbind(dim=1))
        if self.res_link:
            y = self.skip(x, gain=np.sqrt(0.5))
        x = self.conv0(x, _style_in(style_cache, self.conv0, w_iter, w0), fused_modconv=fm, noise_mode=noise_mode, residual=x0,
                       styles_sd=_style_of(style_cache, self.conv0))
        if self.res_link:
            x = self.conv1(x, _style_in(style_cache, self.conv1, w_iter, w0), fused_modconv=fm, gain=np.sqrt(0.5), noise_mode=noise_mode,
                           residual=y, styles_sd=_style_of(style_cache, self.conv1))
        else:
            x = self.conv1(x, _style_in(style_cache, self.conv1, w_iter, w0), fused_modconv=fm, noise_mode=noise_mode,
                           styles_sd=_style_of(style_cache, self.conv1))
        if self.torgb is not None:
            img = self.torgb(x, _style_in(style_cache, self.torgb, w_iter, w0), fused_modconv=fm, base_img=img, base_filter=self.resample_filter,
                             styles_sd=_style_of(style_cache, self.torgb))
        elif img is not None:
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        return x, img


@register('comodgan_synthesis', version)
class Synthesis(nn.Module):
    def __init__(self, w_dim=512, w0_dim=1024, resolution=256, rgb_n=3, ch_base=16384, ch_max=512, use_fp16_after_res=16,
                 resample_filter=[1, 3, 3, 1], activation=_ACT):
        super().__init__()
        log2res = int(np.log2(resolution))
        if 2 ** log2res != resolution:
            raise ValueError
        self.block_res = [2 ** i for i in range(2, log2res + 1)]
        self.w_dim, self.resolution, self.rgb_n = w_dim, resolution, rgb_n
        if resolution in (256, 512, 1024):        # comodgan.py:367-372
            self.num_ws = {256: 14, 512: 16, 1024: 18}[resolution]
        c4 = min(ch_base // self.block_res[0], ch_max)
        self.b4 = synthesis_block_first(w0_dim, c4, w_dim, resolution=4, rgb_n=rgb_n, activation=activation)
        for ri, rj in zip(self.block_res[:-1], self.block_res[1:]):
            ci, cj = min(ch_base // ri, ch_max), min(ch_base // rj, ch_max)
            setattr(self, 'b{}'.format(rj), synthesis_block(ci, cj, w_dim=w_dim, w0_dim=w0_dim, resolution=rj, rgb_n=rgb_n,
                                                          resample_filter=resample_filter, activation=activation, res_link=False,
                                                          use_fp16=(use_fp16_after_res is not None and rj > use_fp16_after_res)))     # comodgan.py:382
        self._any_fp16 = use_fp16_after_res is not None and resolution > use_fp16_after_res

    def forward(self, x, feats, ws, noise_mode='random'):
        ws = ws.to(torch.float32)
        block_ws = []
        w_idx = 0
        for res in self.block_res:   # a block's torgb shares the next block's first w (comodgan.py:399-403)
            block = getattr(self, f'b{res}')
            block_ws.append(ws.narrow(1, w_idx, block.num_conv + block.num_torgb))
            w_idx += block.num_conv
        w0 = x
        # (training rows: every layer derives its own styles under autograd; the grouped style kernels are an inference-path fusion)
        # (... and so is the style cache: with float16 blocks every layer takes the generic route and derives its own styles)
        cache = None if self.__dict__.get('_any_fp16') or grad_ops.wants_grad(x, ws, *self.parameters()) else self._all_styles(ws, w0.to(torch.float32))
        x, img = self.b4(x, feats[4], block_ws[0], noise_mode=noise_mode, style_cache=cache)
        for res, cur_ws in zip(self.block_res[1:], block_ws[1:]):
            x, img = getattr(self, f'b{res}')(x, feats[res], img, cur_ws, w0, noise_mode=noise_mode, style_cache=cache)
        return img

    def _all_styles(self, ws, w0):
        """Every layer's styles = affine(cat[w_i, x_global]) (stylegan.py:284,327), their batch normalisation and the
        demodulation coefficients (stylegan.py:147-155) in two grouped launches instead of three small ones per layer.
        Returns {id(layer): (s [N,I], dcoef [N,O] | None)}."""
        plan = []
        w_idx = 0
        for res in self.block_res:
            block = getattr(self, f'b{res}')
            layers = [block.conv] if res == self.block_res[0] else [block.conv0, block.conv1]
            if block.torgb is not None:
                layers.append(block.torgb)
            for j, layer in enumerate(layers):
                plan.append((layer, w_idx + j))
            w_idx += block.num_conv
        n, dev = ws.shape[0], ws.device
        tot_i = sum(l.affine.weight.shape[0] for l, _ in plan)
        tot_o = sum(l.weight.shape[0] for l, _ in plan if not isinstance(l, torgb_layer))
        raw = torch.empty((n, tot_i), device=dev, dtype=torch.float32)
        s_all = torch.empty(n * tot_i, device=dev, dtype=torch.float32)
        d_all = torch.empty(max(n * tot_o, 1), device=dev, dtype=torch.float32)
        dense_items, prep_items, cache = [], [], {}
        oi = od = 0
        for layer, wi in plan:
            i_n = layer.affine.weight.shape[0]
            aff = layer.affine
            st = raw[:, oi:oi + i_n]
            dense_items.append(dict(x1=ws[:, wi, :], x2=w0, w=aff.weight.detach(), b=None if aff.bias is None else aff.bias.detach(),
                                    y=st, wgain=aff.weight_gain, bgain=aff.bias_gain))
            s = s_all[n * oi:n * (oi + i_n)].view(n, i_n)
            if isinstance(layer, torgb_layer):
                prep_items.append(dict(styles=st, pw=None, demod=False, pre_gain=layer.weight_gain, s=s, d=None))
                cache[id(layer)] = (s, None)
            else:
                o_n = layer.weight.shape[0]
                d = d_all[n * od:n * (od + o_n)].view(n, o_n)
                od += o_n
                prep_items.append(dict(styles=st, pw=layer.prepped(), demod=True, pre_gain=1.0, s=s, d=d))
                cache[id(layer)] = (s, d)
            oi += i_n
        kernels.dense_grouped(dense_items)
        kernels.modconv_style_prep_grouped(prep_items)
        return cache


@register('comodgan_generator', version)
class Generator(Generator_StyleGan):
    """x [N,4,R,R] (mask-0.5, rgb*mask), z [N,512], c [N,0] -> img [N,3,R,R] (comodgan.py:435-481)."""

    def __init__(self, mapping, encoder, synthesis):
        super().__init__(mapping, synthesis)
        self.encoder = encoder if isinstance(encoder, nn.Module) else get_model()(encoder)
        self.ic_n = self.encoder.ic_n

    _warned_training_route = False

    def forward(self, x, z, c, truncation_psi=1, truncation_cutoff=None, noise_mode='random'):
        if not self.training and not Generator._warned_training_route and torch.is_grad_enabled() \
                and any(p.requires_grad for p in self.parameters()):
            # Routes are chosen by grad mode (grad_ops.wants_grad), like autograd itself: an eval() module whose parameters still
            # require grad, called outside torch.no_grad(), records a graph and runs the differentiable (non-fused) kernels.
            import warnings
            warnings.warn('shgan_amd Generator: eval() module called with gradients enabled and parameters that require grad -- this '
                          'takes the differentiable training route (several times slower than the fused inference kernels). Call '
                          '.requires_grad_(False) as the reference eval loop does (shgan_default.py:255) or wrap the call in '
                          'torch.no_grad().', RuntimeWarning, stacklevel=2)
            Generator._warned_training_route = True
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        x, feats = self.encoder(x)
        return self.synthesis(x, feats, ws, noise_mode=noise_mode)
