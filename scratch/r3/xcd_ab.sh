#!/bin/bash
# A/B of the XCD-aware slot mapping on the same box
S='dec1.2 96->96@64|net dec1.0 96+1->96@64|net dec2.0 96+48->96@32|enc1.2 48->48@64|inf48->48@1024'
IFS='|' read -ra SH <<< "$S"
for x in 0 1 0 1; do
  echo "== SPRK_XCD=$x fp32"; SPRK_XCD=$x python scratch/convbench.py "${SH[@]}"
done
for x in 0 1 0 1; do
  echo "== SPRK_XCD=$x bf16"; SPRK_XCD=$x python scratch/convbench.py bf16 "${SH[@]}"
done
