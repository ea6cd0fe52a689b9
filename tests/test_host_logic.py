"""CPU (no GPU): host-side logic of the product -- C-ABI exports, registry, state_dict schema,
constant tables, argument parsing, and the loud-failure contract for CPU tensors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import shgan_amd
from conftest import ROOT, load_golden
from shgan_amd import _lib, kernels
from shgan_amd.model_zoo import comodgan, get_model, get_unit, shgan, stylegan
from shgan_amd.model_zoo.stylegan_utils import upfirdn2d

ACT = 'lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)'


def build_generator(resolution=256, ch_base=32768, ch_max=512, w_dim=512, z_dim=512, w0_dim=1024):
    num_ws = {256: 14, 512: 16, 1024: 18}[resolution]
    mp = comodgan.Mapping(z_dim=z_dim, c_dim=0, w_dim=w_dim, num_ws=num_ws, num_layers=8, activation=ACT,
                          lr_multiplier=0.01, w_avg_beta=0.995)
    enc = shgan.Encoder(resolution=resolution, ic_n=4, oc_n=w0_dim, ch_base=ch_base, ch_max=ch_max, use_fp16_before_res=None,
                        resample_filter=[1, 3, 3, 1], activation=ACT, mbstd_group_size=0, mbstd_c_n=0, c_dim=None,
                        cmap_dim=None, use_dropout=True, has_extra_final_layer=False, shu_channels=32,
                        shu_df_freedom=[2, 3], shu_df_type='piecewise_linear', shu_input_res=64, shu_lowest_res=4,
                        shu_tail_sigma_mult=3, shu_gaussian_at_input_res=False)
    syn = comodgan.Synthesis(w_dim=w_dim, w0_dim=w0_dim, resolution=resolution, rgb_n=3, ch_base=ch_base, ch_max=ch_max,
                             use_fp16_after_res=None, resample_filter=[1, 3, 3, 1], activation=ACT)
    return comodgan.Generator(mp, enc, syn)


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'shgan_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(shg_[a-z0-9_]+)\s*\(', hdr)))
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/shgan_hip.h but not exported'
    assert sorted(_lib.exported_symbols()) == declared      # the ctypes table mirrors the header
    assert _lib.get_lib().shg_abi_version() == _lib.ABI_VERSION


def test_descriptor_structs_match_the_header_layout(tmp_path):
    """The grouped entry points take arrays of plain C structs: the ctypes mirrors must have the layout a C compiler
    gives include/shgan_hip.h (the header is plain C: compiled here with gcc)."""
    import subprocess
    src = tmp_path / 'layout.c'
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "shgan_hip.h"\n'
        'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(shg_dense_group), offsetof(shg_dense_group, y), '
        'offsetof(shg_dense_group, ld1), offsetof(shg_dense_group, wgain), sizeof(shg_style_group), '
        'offsetof(shg_style_group, dcoef), offsetof(shg_style_group, ld), offsetof(shg_style_group, pre_gain)); return 0; }\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    c_layout = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    D, S = _lib.DenseGroup, _lib.StyleGroup
    assert c_layout == [ctypes.sizeof(D), D.y.offset, D.ld1.offset, D.wgain.offset,
                        ctypes.sizeof(S), S.dcoef.offset, S.ld.offset, S.pre_gain.offset]


def test_registry_names_match_reference():
    names = set(get_model().model.keys())
    for n in ('comodgan_mapping', 'shgan_encoder', 'comodgan_encoder', 'comodgan_synthesis', 'comodgan_generator',
              'stylegan2_mapping', 'stylegan2_synthesis', 'stylegan2_generator'):
        assert n in names
    cfg = dict(type='comodgan_mapping', args=dict(z_dim=8, c_dim=0, w_dim=8, num_ws=14))
    assert isinstance(get_model()(cfg), comodgan.Mapping)


def test_state_dict_schema_matches_reference_full_width():
    g = load_golden('generator_full256_stats')
    G = build_generator(256)
    sd = G.state_dict()
    assert sorted(sd.keys()) == list(g['state_dict_keys'])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd.keys())] == list(g['state_dict_shapes'])
    assert sum(p.numel() for p in G.parameters()) == int(g['nparam'])
    assert (G.z_dim, G.c_dim, G.ic_n, G.num_ws, G.img_resolution) == (512, 0, 4, 14, 256)
    with pytest.raises(ValueError):
        comodgan.Encoder(resolution=200, use_fp16_before_res=None, mbstd_c_n=0)


def test_constant_tables_match_reference():
    g = load_golden('shu')
    cw = shgan.make_cweight([2, 3], (64, 33))
    assert np.abs(cw.numpy() - g['cweight_2x3_64x33']).max() < 1e-6
    shu = shgan.SHU(32, 32, [2, 3], 'piecewise_linear', input_res=64, lowest_res=4, tail_sigma_mult=3)
    for r in (4, 8, 16, 32, 64):
        assert np.abs(shu.gaussian_weight_map[r].numpy() - g[f'gauss_{r}']).max() < 1e-7
    assert 'encoder.shu._cw' not in build_generator(256, 2048, 32, 64, 64, 128).state_dict()


def test_get_unit_and_filter_setup():
    act = get_unit()(ACT)()
    assert (act.alpha, act.clamp) == (0.2, 256) and abs(act.gain - 2 ** 0.5) < 1e-12
    assert get_unit()('relu') is torch.nn.ReLU
    g = load_golden('upfirdn2d')
    assert np.array_equal(upfirdn2d.setup_filter([1, 3, 3, 1]).numpy(), g['setup_filter_1331'])
    assert np.allclose(upfirdn2d.setup_filter([1, 2, 3, 4, 5, 4, 3, 2, 1]).numpy(), g['setup_filter_sep'], rtol=1e-7)
    assert np.allclose(upfirdn2d.setup_filter([[1, 2], [3, 4]], flip_filter=True, gain=4).numpy(),
                       g['setup_filter_gain_flip'], rtol=1e-7)
    assert upfirdn2d._parse_padding([1, 2]) == (1, 1, 2, 2)
    with pytest.raises(AssertionError):
        upfirdn2d._parse_scaling(0)


def test_cpu_tensors_fail_loudly_no_fallback():
    x = torch.zeros(1, 1, 8, 8)
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    with pytest.raises(_lib.ShgError):
        upfirdn2d.upfirdn2d(x, f)
    with pytest.raises(_lib.ShgError):
        kernels.dense(torch.zeros(2, 4), torch.zeros(3, 4))


def test_bad_arguments_are_reported_by_the_c_abi():
    lib = _lib.get_lib()
    oh, ow = ctypes.c_int(), ctypes.c_int()
    assert lib.shg_upfirdn2d_out_size(16, 16, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2, ctypes.byref(oh), ctypes.byref(ow)) == 0
    assert (oh.value, ow.value) == (17, 17)
    rc = lib.shg_upfirdn2d_f32(None, None, None, 1, 1, 4, 4, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, None)
    assert rc == -1 and b'null' in lib.shg_last_error()
    rc = lib.shg_conv2d_f32(ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 4, 4, 64, 8, 8, 5, 5, 0, 2, 1, 0,
                            None, None, None, None, 0, 0.0, 0, 0.0, 1.0, -1.0, None, 0, None, 0, None)
    assert rc == -1 and b'3x3' in lib.shg_last_error()
    # split-K planning is a pure host function: tiny spatial grids split along K (ks x output bytes); big ones only
    # ask for the tail-split scratch of the one-workgroup-per-CU kernels (< 256 tile-slices of 128 x 256 accumulators);
    # 64-channel layers (narrow tiles, several workgroups per CU) need nothing
    small = lib.shg_conv2d_workspace_bytes(16, 512, 512, 4, 4, 3, 3, 0, 1, 1)
    assert small > 0 and small % (16 * 512 * 4 * 4 * 4) == 0
    assert lib.shg_conv2d_workspace_bytes(16, 512, 512, 64, 64, 3, 3, 0, 1, 1) == 256 * 128 * 256 * 4
    assert lib.shg_conv2d_workspace_bytes(16, 64, 64, 512, 512, 3, 3, 0, 1, 1) == 0
