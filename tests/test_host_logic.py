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
    from shgan_amd import configs
    return configs.build_generator(resolution, ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim, w0_dim=w0_dim)


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
    # the bicubic sampling of the control grid (shgan.py:116-118), closed form vs F.grid_sample of the reference
    assert np.abs(shgan.make_cweight([2, 3], (64, 33), type='bicubic').numpy() - g['cweight_bicubic_2x3_64x33']).max() < 2e-6
    assert np.abs(shgan.make_cweight([3, 2], (16, 9), type='bicubic').numpy() - g['cweight_bicubic_3x2_16x9']).max() < 2e-6
    with pytest.raises(NotImplementedError):
        shgan.make_cweight([2, 3], (64, 33), type='nearest')
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


def test_fp16_entry_points_validate_their_arguments_without_a_gpu():
    """The fp16 route of the C ABI (include/shgan_hip.h, ABI 24): argument checks are host code and run here -- null pointers, channel
    counts the operand layout cannot take, unsupported geometry; the pure planning functions are exercised as well."""
    lib = _lib.get_lib()
    P = ctypes.c_void_p(16)
    assert lib.shg_conv2d_f16(None, None, None, None, 1, 32, 32, 8, 8, 3, 1, 1, 0, 0, 8, 8, None) == -1 and b'null' in lib.shg_last_error()
    assert lib.shg_conv2d_f16(P, P, None, P, 1, 24, 32, 8, 8, 3, 1, 1, 0, 0, 8, 8, None) == -1 and b'multiple of 32' in lib.shg_last_error()
    assert lib.shg_conv2d_f16(P, P, None, P, 1, 32, 32, 8, 8, 5, 1, 2, 0, 0, 8, 8, None) == -1                  # 5x5 kernel
    assert lib.shg_conv2d_f16(P, P, None, P, 1, 32, 32, 8, 8, 3, 1, 1, 0, 0, 7, 8, None) == -1 and b'extent' in lib.shg_last_error()
    assert lib.shg_conv2d_f16(P, P, None, P, 1, 32, 32, 8, 8, 3, 1, 0, 1, 0, 17, 17, None) == -1                # transposed form is stride 2
    assert lib.shg_conv2d_f16_pack_weight(P, P, 9, 64, 40, None) == -1
    # packed weights: 32-channel blocks rounded up to whole groups of four, [blocks][taps][I/16][64 lanes][8]
    assert lib.shg_conv2d_f16_packed_weight_elems(9, 64, 64) == 4 * 9 * 4 * 512
    assert lib.shg_conv2d_f16_packed_weight_elems(1, 3, 32) == 4 * 1 * 2 * 512
    assert lib.shg_conv2d_f16_needs_clear(16, 16, 0, 33, 33) == 0 and lib.shg_conv2d_f16_needs_clear(16, 16, 1, 33, 33) == 1
    assert lib.shg_conv2d_wgrad_f16(P, P, P, 1, 20, 32, 8, 8, 8, 8, 3, 1, 1, P, 1 << 30, None) == -1 and b'multiples of 8' in lib.shg_last_error()
    assert lib.shg_conv2d_wgrad_f16(P, P, P, 1, 32, 32, 8, 8, 8, 8, 3, 1, 1, P, 16, None) == -1 and b'workspace' in lib.shg_last_error()
    assert lib.shg_conv2d_wgrad_f16(P, P, P, 1, 32, 32, 8, 8, 4, 4, 1, 2, 0, P, 1 << 30, None) == -1              # 1x1 stride 2
    ws = lib.shg_conv2d_wgrad_f16_workspace_bytes(8, 64, 64, 512, 512, 3)
    assert ws == 512 * 9 * 64 * 64 * 4                                                                           # one (o,i) tile -> 512 pixel slices
    assert lib.shg_upfirdn2d_f16(P, P, P, 1, 12, 8, 8, 4, 4, 1, 1, 1, 1, 2, 2, 2, 2, 0, 1.0, None) == -1 and b'multiple of 8' in lib.shg_last_error()
    assert lib.shg_bias_act_f16(P, None, P, 64, 12, 1, 0.2, 1.0, 256.0, None) == -1
    assert lib.shg_bias_act_f16(P, ctypes.c_void_p(8), P, 64, 16, 1, 0.2, 1.0, 256.0, None) == -1 and b'aligned' in lib.shg_last_error()   # float4 operand loads
    assert lib.shg_modtail_f16(P, ctypes.c_void_p(24), None, 0, None, P, 1, 64, 16, 1, 0.2, 1.0, 256.0, None) == -1 and b'aligned' in lib.shg_last_error()
    assert lib.shg_modtail_backward_f16(P, P, None, None, None, None, P, None, None, 1, 64, 24, 1, 0.2, 1.0, 256.0, None) == -1 and b'power of two' in lib.shg_last_error()
    assert lib.shg_modtail_backward_f16_blocks(64 * 64, 64) == 128 and lib.shg_modtail_backward_f16_blocks(512 * 512, 64) == 256
    assert lib.shg_modtail_backward_f32(P, P, None, None, None, None, P, None, None, 1, 64, 30, 1, 0.2, 1.0, 256.0, None) == -1
    assert lib.shg_modtail_backward_f32_cslices(8, 512, 16) == 64 and lib.shg_modtail_backward_f32_cslices(8, 64, 512 * 512) == 1
    # and the Python wrappers refuse CPU tensors (no fallback)
    from shgan_amd import kernels_f16
    with pytest.raises(_lib.ShgError):
        kernels_f16.conv2d(torch.zeros(1, 32, 8, 8, dtype=torch.float16), torch.zeros(32, 32, 3, 3, dtype=torch.float16))
    with pytest.raises(_lib.ShgError):
        kernels_f16.upfirdn2d(torch.zeros(1, 8, 8, 8, dtype=torch.float16), torch.ones(4, 4))


def test_round5_entry_points_validate_their_arguments_without_a_gpu():
    """The entry points added with ABI 29-35 (Winograd-domain weight gradient, the fp16 route switch, the two glue kernels): geometry
    gates, workspace planning and argument checks are host code -- exercised here without a device."""
    lib = _lib.get_lib()
    P = ctypes.c_void_p(16)
    sup = lib.shg_conv2d_wgrad_wino_supported
    assert sup(64, 64, 64, 64, 3, 3, 1, 1) == 1 and sup(4, 4, 4, 4, 3, 3, 1, 1) == 1
    assert sup(64, 64, 32, 32, 3, 3, 2, 1) == 0                     # stride 2: the direct kernel
    assert sup(64, 64, 64, 64, 1, 1, 1, 0) == 0                     # 1x1
    assert sup(64, 62, 64, 62, 3, 3, 1, 1) == 0                     # rows of whole 16-byte pieces only
    assert sup(64, 64, 62, 62, 3, 3, 1, 0) == 0                     # pad 0
    # workspace = K-slices x [O, I, 3, 3] floats when the (o, i) grid alone cannot fill 256 CUs; none when it does or when there is one chunk
    wsb = lib.shg_conv2d_wgrad_wino_workspace_bytes
    assert wsb(8, 64, 64, 512, 512) % (64 * 64 * 9 * 4) == 0 and wsb(8, 64, 64, 512, 512) // (64 * 64 * 9 * 4) == 128      # 2 blocks -> 128 slices
    assert wsb(8, 512, 512, 64, 64) // (512 * 512 * 9 * 4) == 2                                                              # 128 blocks -> 2 slices
    assert wsb(1, 512, 512, 4, 4) == 0                                                                                        # one chunk of tiles
    f = lib.shg_conv2d_wgrad_wino_f32
    assert f(None, P, P, 1, 32, 32, 16, 16, None, 0, None) == -1 and b'null' in lib.shg_last_error()
    assert f(P, P, P, 1, 32, 32, 16, 18, None, 0, None) == -1 and b'W %' in lib.shg_last_error()
    assert f(ctypes.c_void_p(8), P, P, 1, 32, 32, 16, 16, None, 0, None) == -1 and b'aligned' in lib.shg_last_error()
    assert f(P, P, P, 8, 64, 64, 512, 512, P, 16, None) == -1 and b'workspace' in lib.shg_last_error()
    assert f(P, P, P, 1, 4096, 32, 512, 512, P, 1 << 40, None) == -1 and b'2 GiB' in lib.shg_last_error()
    # fp16 route switch: returns the previous mask, keeps three bits
    old = lib.shg_conv2d_f16_set_routes(0)
    try:
        assert old == 7 and lib.shg_conv2d_f16_set_routes(0xFF) == 0 and lib.shg_conv2d_f16_set_routes(5) == 7
    finally:
        lib.shg_conv2d_f16_set_routes(old)
    assert lib.shg_conv2d_f16_set_routes(old) == old
    # glue kernels
    assert lib.shg_sum_partials_f32(None, P, 1, 4, 64, None) == -1 and lib.shg_sum_partials_f32(P, P, 70000, 4, 64, None) == -1
    assert lib.shg_sum_partials_f32(P, P, 1, 0, 64, None) == -1
    assert lib.shg_scale_cast_f32_f16(None, P, 16, 1.0, 1, None) == -1 and lib.shg_scale_cast_f32_f16(P, P, -1, 1.0, 1, None) == -1
    assert lib.shg_scale_cast_f32_f16(P, P, 0, 1.0, 1, None) == 0                      # empty: nothing launched
    # Winograd launches that do not fill the chip split along the input channels (ABI 35): scratch = slices x output
    out = lambda n, o, h, w: n * o * h * w * 4                                   # noqa: E731
    wsw, wsw4 = lib.shg_conv2d_wino_workspace_bytes, lib.shg_conv2d_wino4_workspace_bytes
    assert wsw(8, 512, 512, 512, 16, 16) == 4 * out(8, 512, 16, 16)            # 64 workgroups -> 4 slices of 16 chunks
    assert wsw(4, 512, 512, 512, 16, 16) == 8 * out(4, 512, 16, 16)            # 32 -> 8 slices of 8 chunks (no thinner)
    assert wsw(16, 64, 64, 64, 16, 16) == 0                                     # 8 chunks: nothing to split
    assert wsw4(8, 512, 512, 512, 32, 32) == 2 * out(8, 512, 32, 32) and wsw4(16, 512, 512, 512, 32, 32) == 0
    assert wsw4(16, 64, 64, 64, 512, 512) == 0
    wsu = lib.shg_conv2d_up_poly_workspace_bytes
    assert wsu(4, 512, 512, 512, 16, 16) == 4 * (4 * out(4, 512, 17, 17)) and wsu(16, 512, 512, 512, 16, 16) == 0 and wsu(4, 512, 512, 512, 16, 18) == 0
    P2 = ctypes.c_void_p(32)
    assert lib.shg_conv2d_wino_ws_f32(P2, P2, P2, 1, 64, 64, 64, 16, 18, None, None, None, None, 0, 0.0, 0, 0.2, 1.0, -1.0, None, None, 0, None) == -1 \
        and b'W %' in lib.shg_last_error()
    # the native op over the plugin's whole operand range (ABI 34): dtype code, strides, the size rule of upfirdn2d.cpp:26-36
    st = (ctypes.c_long * 4)(96, 16, 4, 1)
    ufs = lib.shg_upfirdn2d_strided
    assert ufs(P, P, P, 3, 1, 6, 4, 4, st, st, 4, 4, 4, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, None) == -1 and b'dtype' in lib.shg_last_error()
    assert ufs(P, P, P, 2, 1, 6, 4, 4, None, st, 4, 4, 4, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, None) == -1 and b'null' in lib.shg_last_error()
    assert ufs(P, P, P, 2, 1, 6, 4, 4, st, st, 4, 4, 4, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, None) == -1 and b'upsampling' in lib.shg_last_error()
    assert ufs(P, P, P, 0, 1, 6, 2, 2, st, st, 4, 4, 4, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, None) == -1 and b'at least 1x1' in lib.shg_last_error()
    neg = (ctypes.c_long * 4)(96, 16, -4, 1)
    assert ufs(P, P, P, 0, 1, 6, 4, 4, neg, st, 2, 2, 2, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, None) == -1 and b'stride' in lib.shg_last_error()
    with pytest.raises(_lib.ShgError):
        kernels.upfirdn2d_strided(torch.zeros(1, 2, 4, 4, dtype=torch.float64), torch.ones(2, 2))
    with pytest.raises(_lib.ShgError):
        kernels.sum_partials(torch.zeros(2, 4, 64))
    with pytest.raises(_lib.ShgError):
        kernels.scale_cast(torch.zeros(8), 1.0, True)
    with pytest.raises(_lib.ShgError):
        kernels.conv2d_wgrad(torch.zeros(1, 32, 16, 16), torch.zeros(1, 32, 16, 16), 3, 3, 1, 1)


def test_configs_build_and_seeded_init_are_deterministic():
    """The product constructs the shipped generators by itself (registry configs = the flattened YAML of SURVEY A.1) and
    initialises them identically in every process: same seed -> bit-identical state dict, reference initialiser statistics."""
    from shgan_amd import configs
    kw = dict(ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    cfg = configs.model_cfg('shgan_g256', **kw)
    assert cfg['type'] == 'comodgan_generator' and cfg['args']['encoder']['type'] == 'shgan_encoder'
    assert cfg['args']['encoder']['args']['shu_df_freedom'] == [2, 3] and cfg['args']['mapping']['args']['num_ws'] == 14
    assert configs.model_cfg('shgan_g512')['args']['synthesis']['args']['resolution'] == 512
    c1024 = configs.model_cfg('shgan_g1024')                                    # configs/model/shgan.yaml:94-124
    assert c1024['args']['mapping']['args']['num_ws'] == 18 and c1024['args']['encoder']['args']['resolution'] == 1024
    assert c1024['args']['encoder']['args']['shu_input_res'] == 64 and c1024['args']['synthesis']['args']['resolution'] == 1024
    with pytest.raises(KeyError):
        configs.model_cfg('shgan_g128')
    g1024 = configs.build_generator(1024, **kw)
    assert g1024.num_ws == 18 and 'synthesis.b1024.torgb.weight' in g1024.state_dict() and 'encoder.b1024.fromrgb.weight' in g1024.state_dict()
    a = configs.seeded_init_(configs.build_generator(256, **kw), seed=3, noise_strength=0.1, bias_std=0.1).state_dict()
    b = configs.seeded_init_(configs.build_generator(256, **kw), seed=3, noise_strength=0.1, bias_std=0.1).state_dict()
    c = configs.seeded_init_(configs.build_generator(256, **kw), seed=4).state_dict()
    assert list(a.keys()) == list(b.keys()) and all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a['synthesis.b8.conv0.weight'], c['synthesis.b8.conv0.weight'])
    assert float(c['synthesis.b8.conv0.noise_strength']) == 0.0 and float(a['synthesis.b8.conv0.noise_strength']) == pytest.approx(0.1)
    assert torch.all(c['synthesis.b8.conv0.affine.bias'] == 1) and torch.all(c['synthesis.b8.conv0.bias'] == 0)
    assert abs(float(c['mapping.fc3.weight'].std()) - 100.0) < 5.0                     # randn / lr_multiplier
    assert abs(float(c['encoder.shu.df1.weight'].mean()) - 1 / 64) < 1e-3            # N(1/C, 0.1/C)
    assert abs(float(c['encoder.shu.conv0.weight'].std()) - 1 / 8) < 0.01            # He-normal, fan-in 64


def test_activation_arguments_reach_the_kernels():
    """``dense`` and the thin 1x1 convolution pass the lrelu_agc instance's alpha / gain / clamp on, they do
    not fall back to the defaults of the wrapper."""
    from shgan_amd.model_zoo import stylegan
    seen = {}

    def fake_dense(x, w, b=None, **kw):
        seen['dense'] = kw
        return x

    def fake_thin(x, w, b=None, **kw):
        seen['thin'] = kw
        return x
    old = kernels.dense, kernels.conv1x1_thin_in
    kernels.dense, kernels.conv1x1_thin_in = fake_dense, fake_thin
    try:
        d = stylegan.dense(8, 8, activation='lrelu_agc(alpha=0.1, gain=1)')
        d(torch.zeros(2, 8))
        assert seen['dense']['act'] is True and seen['dense']['alpha'] == 0.1 and seen['dense']['act_gain'] == 1
        assert seen['dense']['clamp'] is None
        layer = stylegan.conv2d_layer(4, 16, 1, activation='lrelu_agc(alpha=0.3, gain=sqrt_2, clamp=7)')
        layer(torch.zeros(1, 4, 8, 8), gain=0.5)
        assert seen['thin']['alpha'] == 0.3 and seen['thin']['clamp'] == 7 and seen['thin']['gain'] == 0.5
    finally:
        kernels.dense, kernels.conv1x1_thin_in = old


def test_channel_sum_and_channel_bias_node_on_cpu():
    """``grad_ops.channel_sum`` (the two-step form for thin tensors) equals the plain reduction, and ``add_channel_bias`` has the gradients
    of ``y + bias.view(1, -1, 1, 1)`` -- first and second order (pure tensor code: runs without the HIP library)."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo.stylegan_utils import grad_ops
    g = torch.Generator().manual_seed(3)
    for shape in ((2, 3, 128, 128), (2, 3, 8, 8), (1, 40, 128, 130)):
        t = torch.randn(shape, generator=g, dtype=torch.float64)
        assert torch.allclose(grad_ops.channel_sum(t.float()).double(), t.sum([0, 2, 3]), rtol=1e-5, atol=1e-4)
    w = torch.randn(2, 3, 130, 128, generator=g, dtype=torch.float64)
    res = []
    with torch.enable_grad():
        for fn in (grad_ops.add_channel_bias, lambda yy, bb: yy + bb.view(1, -1, 1, 1)):
            y = torch.randn(2, 3, 130, 128, generator=torch.Generator().manual_seed(4), dtype=torch.float64, requires_grad=True)
            b = torch.randn(3, generator=torch.Generator().manual_seed(5), dtype=torch.float64, requires_grad=True)
            out = fn(y, b)
            gy, gb = torch.autograd.grad((out.square() * w).sum(), [y, b], create_graph=True)
            ggy, ggb = torch.autograd.grad(gb.square().sum() + gy.sum(), [y, b])
            res.append((out.detach(), gy.detach(), gb.detach(), ggy, ggb))
    for got, want in zip(*res):
        assert torch.allclose(got.double(), want.double(), rtol=1e-6, atol=1e-6)


def test_bench_reads_its_committed_profiles():
    """bench.py attaches constants of committed profiles to its JSON line (PMC traffic per kernel class, the SHU floor probe): the parsers
    must find what the files under profiles/ hold (a renamed file or a reformatted table would silently drop the fields)."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location('shg_bench_for_test', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fl = bench.shu_floors()
    assert set(fl) == {'shu_rfft2', 'shu_spectral', 'shu_irfft2'}
    for product, skeleton, empty in fl.values():
        assert product > skeleton > empty > 0
    gb, src = bench.pmc_traffic(512, 16)
    assert gb and gb > 0.5 and src == os.path.join('profiles', 'traffic_512x16.json')
    cls = bench.pmc_traffic_classes(512, 16)
    assert 'conv_wino4_kernel' in cls and cls['conv_wino4_kernel']['read_GB_per_launch'] > 0
