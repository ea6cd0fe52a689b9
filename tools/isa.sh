#!/bin/bash
# compile one csrc/*.hip with the product flags, keep the ISA in /tmp, print the resource summary: tools/isa.sh conv_f16_ring [extra -D flags]
F=$1; shift
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -Wno-inline-asm -Wno-unused-value "$@" \
  -c /root/repo/sh-gan_amd/csrc/$F.hip -o /tmp/$F.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E 'Function Name|VGPRs|AGPRs|Spill|Occupancy|error|ScratchSize' | grep -v 'SGPRs:'
echo ISA: /tmp/$F-hip-amdgcn-amd-amdhsa-gfx950.s
