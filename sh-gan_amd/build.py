"""Build recipe for libshgan_hip.so (gfx950 only, hipcc, in-tree output).

``python sh-gan_amd/build.py`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
The shared object is linked WITHOUT an rpath to /opt/rocm so that, inside a PyTorch-ROCm process,
it binds to the HIP runtime torch has already loaded (one runtime per process)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIBNAME = 'libshgan_hip.so'
VARDIR = os.path.join(os.path.dirname(HERE), 'tools', '_variants')      # study builds (-DSHG_ABLATE, A/B knobs) live with the tools, never beside the product library
SOURCES = ['capi.hip', 'upfirdn2d.hip', 'pointwise.hip', 'dense.hip', 'conv_mfma.hip', 'conv_wino.hip', 'conv_wino4.hip', 'conv_wino_poly.hip', 'conv_wgrad.hip', 'conv_wgrad_wino.hip', 'conv_f16.hip', 'conv_f16_ring.hip', 'conv_f16_upring.hip', 'conv_f16_down.hip', 'shu.hip', 'mask_raster.hip', 'fid_stats.hip']
# per-source extras: the Winograd weight-gradient transforms are scalar fp32 chains beside MFMAs -- SLP-packed (v_pk_*) forms cost register
# moves and issue slots there
SRC_FLAGS = {'conv_wgrad_wino.hip': ['-fno-slp-vectorize']}
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function', '-Wno-inline-asm']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


INCLUDE = os.path.join(os.path.dirname(HERE), 'include')


def _digest(extra=()):
    """sha256 over every kernel source, the public C header (an ABI struct edit must trigger a rebuild) and the flags."""
    h = hashlib.sha256()
    for d in (CSRC, INCLUDE):
        for name in sorted(os.listdir(d)):
            with open(os.path.join(d, name), 'rb') as fh:
                h.update(name.encode())
                h.update(fh.read())
    h.update(' '.join(list(FLAGS) + [f'{k}:{v}' for k, v in sorted(SRC_FLAGS.items())] + list(extra)).encode())
    return h.hexdigest()


def lib_path(ablate=False, variant=None):
    if variant:
        return os.path.join(VARDIR, f'libshgan_hip_{variant}.so')
    return os.path.join(VARDIR, 'libshgan_hip_ablate.so') if ablate else os.path.join(LIBDIR, LIBNAME)


def build(force=False, verbose=True, ablate=False, variant=None, defines=()):
    """``ablate=True`` builds the timing-study variant (-DSHG_ABLATE: the SHG_*_DBG / SHG_CONV_VARIANT environment switches
    that make kernels skip work) as a SEPARATE library for tools/; the product library has no such switches.
    ``variant='name', defines=['-DX=1']`` builds libshgan_hip_<name>.so with extra compile-time knobs for A/B runs in tools/."""
    os.makedirs(LIBDIR, exist_ok=True)
    outdir = VARDIR if (ablate or variant) else LIBDIR
    os.makedirs(outdir, exist_ok=True)
    extra = (['-DSHG_ABLATE'] if ablate else []) + list(defines)
    if variant:
        ablate = True            # (shares the .abl.o object names / separate stamp below)
    stamp = os.path.join(outdir, f'.build_digest_{variant}' if variant else ('.build_digest_ablate' if ablate else '.build_digest'))
    dig = _digest(extra)
    if not force and os.path.exists(lib_path(ablate, variant)) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        if verbose:
            print(f'[build] {os.path.basename(lib_path(ablate, variant))} up to date')
        return lib_path(ablate, variant)
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(outdir, src.replace('.hip', (f'.{variant}.o' if variant else '.abl.o') if ablate else '.o'))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + SRC_FLAGS.get(src, []) + extra + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print('[build]', ' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode(errors='replace'))
            raise RuntimeError(f'hipcc failed on {src}')
        if verbose and out.strip():
            print(out.decode(errors='replace'))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path(ablate, variant)] + objs
    if verbose:
        print('[build]', ' '.join(cmd))
    subprocess.check_call(cmd)
    for obj in objs:
        os.remove(obj)
    with open(stamp, 'w') as fh:
        fh.write(dig)
    return lib_path(ablate, variant)


if __name__ == '__main__':
    var = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--variant=')]
    build(force='--force' in sys.argv, ablate='--ablate' in sys.argv, variant=var[0] if var else None,
          defines=[a for a in sys.argv[1:] if a.startswith('-D')])
