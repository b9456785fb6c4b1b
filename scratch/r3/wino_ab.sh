#!/bin/bash
for rep in 1 2; do
for v in 0 1 2 3; do
  echo "== variant $v"; SPRK_LIB=$PWD/scratch/r3/libsprk_v$v.so python scratch/convbench.py "dec1.2 96->96@64" "enc1.2 48->48@64" "net dec2.0 96+48->96@32" 2>&1 | grep -v amdgpu.ids
done
done
