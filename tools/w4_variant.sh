#!/bin/bash
# Build libshgan_hip_<tag>.so that differs from the product library only in conv_wino4.hip compiled with extra -D knobs
# (A/B timing runs: SHG_VARIANT=<tag> python tools/conv_bench.py).  usage: [W4SRC=<other conv_wino4.hip>] tools/w4_variant.sh <tag> [-DKNOB=1 ...]
# W4SRC: compile that file instead of csrc/conv_wino4.hip (e.g. `git show <rev>:sh-gan_amd/csrc/conv_wino4.hip > /tmp/old.hip`) -- box-to-box
# clock differences are +-5 %, so an old and a new kernel are only comparable inside ONE gpurun call.
set -e
cd "$(dirname "$0")/../sh-gan_amd"
V=../tools/_variants
TAG=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-inline-asm"
mkdir -p $V/objcache
for f in csrc/*.hip; do
  b=$(basename $f .hip); [ $b = conv_wino4 ] && continue
  if [ ! -f $V/objcache/$b.o ] || [ $f -nt $V/objcache/$b.o ]; then /opt/rocm/bin/hipcc $FLAGS -c $f -o $V/objcache/$b.o & fi
done
wait
/opt/rocm/bin/hipcc $FLAGS -Icsrc "$@" -c ${W4SRC:-csrc/conv_wino4.hip} -o $V/objcache/conv_wino4.$TAG.obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libshgan_hip_$TAG.so $V/objcache/*.o $V/objcache/conv_wino4.$TAG.obj
echo built $V/libshgan_hip_$TAG.so
