#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "16bit" > $O/r2_c16_tests.log 2>&1; tail -6 $O/r2_c16_tests.log | cut -c1-400
timeout -k 10 300 python scratch/convbench.py bf16 "head 384->384 1x1@64" "head 384->96 1x1@64" 2>&1 | grep -v amdgpu
SPRK_CONV16_HEAD=0 timeout -k 10 300 python scratch/convbench.py bf16 "head 384->384 1x1@64" 2>&1 | grep -v amdgpu
