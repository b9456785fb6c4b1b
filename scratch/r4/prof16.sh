#!/bin/bash
# kernel trace of the eager bf16 training step: bash scratch/r4/prof16.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/p16_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/db -o t -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --infer-size 0 --infer-large 0 --also-dtype none --graph off --sustain-seconds 0 --batch16 off --full-pipeline off --event-steps 0 --dtype bf16 "$@" > $OUT/run.log 2>&1
python3 $R/scratch/r4/kstat.py $OUT/db 45 > $OUT/kstat.txt 2>&1
rm -rf $OUT/db
tail -2 $OUT/run.log | cut -c1-200
