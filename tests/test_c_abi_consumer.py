"""The drop-in boundary from the other side: a plain-C program (tests/c_abi/consumer.c -- C99, gcc, no Python, no PyTorch) includes
include/shgan_hip.h, links libshgan_hip.so + the HIP runtime and checks the kernels against the plain-C oracle."""
import os
import subprocess

import pytest

from conftest import ROOT

BUILD = os.path.join(ROOT, 'tests', 'c_abi', '_build')
EXE = os.path.join(BUILD, 'consumer')
LIBDIR = os.path.join(ROOT, 'sh-gan_amd', 'lib')


def _build():
    import shgan_amd  # noqa: F401
    from shgan_amd import build as shg_build
    shg_build.build(verbose=False)
    os.makedirs(BUILD, exist_ok=True)
    cmd = ['gcc', '-std=c99', '-O1', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'tests', 'c_abi', 'consumer.c'), os.path.join(ROOT, 'oracle', 'native_oracle.c'),
           '-L' + LIBDIR, '-lshgan_hip', '-L/opt/rocm/lib', '-lamdhip64', '-lm', '-o', EXE]
    subprocess.check_call(cmd)


def test_header_is_plain_c_and_the_library_links_without_python():
    """C99 compile of the public header + link against the shared library (no compute: runs without a GPU)."""
    _build()
    assert os.path.exists(EXE)
    out = subprocess.run(['ldd', EXE], stdout=subprocess.PIPE, env=dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ':/opt/rocm/lib')).stdout.decode()
    assert 'libshgan_hip.so' in out and 'libtorch' not in out and 'libpython' not in out


@pytest.mark.gpu
def test_c_consumer_runs_the_kernels_on_the_gpu():
    _build()
    r = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                       env=dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ':/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', '')))
    out = r.stdout.decode()
    assert r.returncode == 0 and 'consumer ok' in out, out
