#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -m gpu -q -x -k "16bit or tiled or translation" > $O/r2_c16_tests.log 2>&1; tail -4 $O/r2_c16_tests.log | cut -c1-400
S='"dec1.2 96->96@64" "net dec1.0 96+1->96@64" "net dec2.0 96+48->96@32" "enc1.2 48->48@64" "net dec3.0 96+48->96@16" "enc48->48@32"'
eval timeout -k 10 300 python scratch/convbench.py bf16 $S 2>&1 | grep -v amdgpu
echo "--- one tile per workgroup"
SPRK_C16_PERSIST=0 timeout -k 10 300 python scratch/convbench.py bf16 "dec1.2 96->96@64" "enc1.2 48->48@64" 2>&1 | grep -v amdgpu
