#!/bin/bash
# scratch/r3/stamp_run.sh <variant .so number (built with -DWINO_STAMP)>
V=$1
rm -f gpurun_out/stamps_v$V.txt
SPRK_WINO_STAMP=$PWD/gpurun_out/stamps_v$V.txt SPRK_LIB=$PWD/scratch/r3/libsprk_v$V.so python scratch/convbench.py "dec1.2 96->96@64" 2>&1 | grep -v amdgpu.ids
python scratch/r3/stamps.py gpurun_out/stamps_v$V.txt
