#!/bin/bash
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 4 32 64; do
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVES -d $R/gpurun_out/pmc_c$c -o p -- python3 $R/scratch/convbench.py "cin$c->96@64" > $R/gpurun_out/pmc_c$c.log 2>&1
done
