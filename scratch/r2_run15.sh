#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
S='"dec1.2 96->96@64" "net dec2.0 96+48->96@32" "enc1.2 48->48@64" "head 384->384 1x1@64" "head 384->96 1x1@64"'
eval timeout -k 10 300 python scratch/convbench.py bf16 $S 2>&1 | grep -v amdgpu
eval timeout -k 10 300 python scratch/convbench.py f32 $S 2>&1 | grep -v amdgpu
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > $O/r2_ops_tests.log 2>&1; tail -3 $O/r2_ops_tests.log | cut -c1-300
