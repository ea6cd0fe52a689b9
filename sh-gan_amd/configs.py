"""Resolved model configurations of the SH-GAN generators (the YAML inheritance of the reference's
configs/model/{stylegan,comodgan,shgan}.yaml flattened: SURVEY.md appendix A.1) and a deterministic
random initialiser, so that the product can construct ``shgan_g256`` / ``shgan_g512`` / ``shgan_g1024`` by itself:

    G = configs.build_generator(512)            # == get_model()(configs.model_cfg('shgan_g512'))
    configs.seeded_init_(G, seed=0)             # the reference's initialisers, identical on every rank

``seeded_init_`` replaces the DDP-constructor weight broadcast of the reference
(lib/experiments/shgan_default.py:231): every rank draws the same numbers from a seeded CPU generator,
so no 317 MB broadcast is needed before a batch-sharded run."""
import math

import torch

ACT = 'lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)'
NUM_WS = {256: 14, 512: 16, 1024: 18}                    # comodgan.py:367-372


def model_cfg(name='shgan_g512', ch_base=32768, ch_max=512, w_dim=512, z_dim=512, w0_dim=1024, use_fp16_before_res=None,
              use_fp16_after_res=None):
    """Registry config (``type`` / ``args``) of a shipped generator; the width arguments exist for reduced-size tests.  The shipped
    configs are float32 (``use_fp16_*: null``, comodgan.yaml:27,46); ``use_fp16_before_res`` (encoder blocks above that resolution) /
    ``use_fp16_after_res`` (synthesis blocks above it) switch the reference's half-precision blocks on (BASELINE config 5)."""
    if name not in ('shgan_g256', 'shgan_g512', 'shgan_g1024'):            # configs/model/shgan.yaml:51-124
        raise KeyError(f'unknown model config {name!r} (shipped: shgan_g256, shgan_g512, shgan_g1024)')
    res = int(name.split('_g')[1])
    mapping = dict(type='comodgan_mapping', args=dict(
        z_dim=z_dim, c_dim=0, w_dim=w_dim, num_ws=NUM_WS[res], num_layers=8, embed_features=None, layer_features=None,
        activation=ACT, lr_multiplier=0.01, w_avg_beta=0.995))
    encoder = dict(type='shgan_encoder', args=dict(
        resolution=res, ic_n=4, oc_n=w0_dim, ch_base=ch_base, ch_max=ch_max, use_fp16_before_res=use_fp16_before_res,
        resample_filter=[1, 3, 3, 1], activation=ACT, mbstd_group_size=0, mbstd_c_n=0, c_dim=None, cmap_dim=None,
        use_dropout=True, has_extra_final_layer=False, shu_channels=32, shu_df_freedom=[2, 3],
        shu_df_type='piecewise_linear', shu_input_res=64, shu_lowest_res=4, shu_tail_sigma_mult=3,
        shu_gaussian_at_input_res=False))
    synthesis = dict(type='comodgan_synthesis', args=dict(
        w_dim=w_dim, w0_dim=w0_dim, resolution=res, rgb_n=3, ch_base=ch_base, ch_max=ch_max, use_fp16_after_res=use_fp16_after_res,
        resample_filter=[1, 3, 3, 1], activation=ACT))
    return dict(type='comodgan_generator', args=dict(mapping=mapping, encoder=encoder, synthesis=synthesis))


def discriminator_cfg(resolution=512, ch_base=32768, ch_max=512, use_fp16_before_res=None):
    """comodgan.yaml:51-58 / stylegan.yaml:31-46: the training-time critic (mask + image, 4 input channels)."""
    return dict(type='stylegan2_discriminator', args=dict(
        resolution=resolution, ic_n=4, ch_base=ch_base, ch_max=ch_max, use_fp16_before_res=use_fp16_before_res,
        resample_filter=[1, 3, 3, 1], activation=ACT, mbstd_group_size=4, mbstd_c_n=1, c_dim=None, cmap_dim=None))


def build_generator(resolution=512, **widths):
    from .model_zoo import get_model
    return get_model()(model_cfg(f'shgan_g{resolution}', **widths))


@torch.no_grad()
def seeded_init_(model, seed=0, noise_strength=0.0, bias_std=0.0):
    """Re-draw every parameter / random buffer of ``model`` from ``torch.Generator(seed)`` on the CPU with the
    reference's initialisers (stylegan.py:40-49,80,219,266,270-271; shgan.py:275): N(0,1) weights (``dense`` weights
    divided by their lr multiplier), zero biases (``bias_init`` of the affine layers: 1), N(0,1) ``noise_const``, zero
    ``noise_strength``, He-normal SHU conv0, N(1/C, 0.1/C) heterogeneous-filter weights.  ``noise_strength`` /
    ``bias_std`` optionally perturb the zero-initialised entries (parity tests use this to exercise those data paths).
    Draws follow the module tree order, so two processes with the same seed hold identical weights."""
    from .model_zoo import shgan, stylegan
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))

    def rn(shape):
        return torch.randn(tuple(shape), generator=g, dtype=torch.float32)

    def put(t, v):
        t.copy_(v.to(device=t.device, dtype=t.dtype))

    def zero_or_noise(t):
        put(t, rn(t.shape) * bias_std if bias_std else torch.zeros(t.shape))

    for _, m in model.named_modules():
        if isinstance(m, stylegan.dense):
            put(m.weight, rn(m.weight.shape) / m.lr_multi)
            if m.bias is not None:
                if m.bias_init:
                    put(m.bias, torch.full(m.bias.shape, float(m.bias_init)))
                else:
                    zero_or_noise(m.bias)
        elif isinstance(m, stylegan.conv2d):                       # SHU conv0: He-normal, use_wscale=False
            fan = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            put(m.weight, rn(m.weight.shape) / math.sqrt(fan))
            if m.bias is not None:
                zero_or_noise(m.bias)
        elif isinstance(m, shgan.heterogeneous_filter):
            c = m.weight.shape[0]
            put(m.weight, 1.0 / c + rn(m.weight.shape) * (0.1 / c))
        elif isinstance(m, stylegan.conv2d_layer):                 # incl. synthesis_layer / torgb_layer
            put(m.weight, rn(m.weight.shape))
            if m.bias is not None:
                zero_or_noise(m.bias)
            if getattr(m, 'use_noise', False):
                put(m.noise_const, rn(m.noise_const.shape))
                put(m.noise_strength, torch.full((), float(noise_strength)))
        elif isinstance(m, stylegan.Mapping) and hasattr(m, 'w_avg'):
            m.w_avg.zero_()
        const = m.__dict__.get('_parameters', {}).get('const')
        if const is not None:
            put(const, rn(const.shape))
    return model
