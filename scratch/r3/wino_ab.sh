#!/bin/bash
# same-box A/B of wino_conv_kernel builds: scratch/r3/libsprk_v<N>.so (built with -DWINO_VARIANT=N; v6 = -fno-slp-vectorize)
for rep in 1 2; do
for v in "$@"; do
  echo "== variant $v"; SPRK_LIB=$PWD/scratch/r3/libsprk_v$v.so python scratch/convbench.py "dec1.2 96->96@64" "net dec2.0 96+48->96@32" "inf48->48@1024" 2>&1 | grep -v amdgpu.ids
done
done
