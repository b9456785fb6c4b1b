#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "16bit" > $O/r2_c16_tests.log 2>&1; tail -8 $O/r2_c16_tests.log | cut -c1-300
bash scratch/pmc16.sh c16a bf16 "dec1.2 96->96@64"
