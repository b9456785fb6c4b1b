#!/bin/bash
# usage: scratch/pmc16.sh <tag> <dtype> <shape name>   -> gpurun_out/pmc_<tag>_{a,b}/ ; prints the counters
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc_$1_a -o p -- python3 $R/scratch/convbench.py $2 "$3" > $R/gpurun_out/pmc_$1_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/pmc_$1_b -o p -- python3 $R/scratch/convbench.py $2 "$3" > $R/gpurun_out/pmc_$1_b.log 2>&1
tail -1 $R/gpurun_out/pmc_$1_a.log
python3 $R/scratch/pmc_read.py $(ls $R/gpurun_out/pmc_$1_a/*/*.db $R/gpurun_out/pmc_$1_a/*.db 2>/dev/null | head -1) $(ls $R/gpurun_out/pmc_$1_b/*/*.db $R/gpurun_out/pmc_$1_b/*.db 2>/dev/null | head -1) > $R/gpurun_out/pmc_$1.txt 2>&1
rm -rf $R/gpurun_out/pmc_$1_a $R/gpurun_out/pmc_$1_b
cat $R/gpurun_out/pmc_$1.txt
