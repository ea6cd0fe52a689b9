#!/bin/bash
# usage: tools/w4_asm.sh <out-prefix> [kernel-regex] [-D...]   -- device assembly + resource usage of conv_wino4.hip
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
KRE=${1:-conv_wino4}; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-inline-asm"
/opt/rocm/bin/hipcc $F "$@" -S --cuda-device-only -o $OUT.s $R/sh-gan_amd/csrc/conv_wino4.hip 2>&1 | grep -v "hip-link"
/opt/rocm/bin/hipcc $F "$@" -Rpass-analysis=kernel-resource-usage -c $R/sh-gan_amd/csrc/conv_wino4.hip -o $OUT.o 2>&1 | grep -E "error|Function Name|VGPRs:|TotalSGPRs|Spill|Scratch|LDS Size" | sed 's/.*remark: *//'
