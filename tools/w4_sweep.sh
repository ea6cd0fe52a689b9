#!/bin/bash
# usage (on the GPU box): tools/w4_sweep.sh tag1 tag2 ...   -- times the F(4x4) kernel of every variant library
python tools/wino4_check.py 2>&1 | grep "F(4x4)" | sed "s/^/product /"
for t in "$@"; do SHG_VARIANT=$t python tools/wino4_check.py 2>&1 | grep "F(4x4)" | awk -v t=$t '{print t, $1, $8, $9}'; done
