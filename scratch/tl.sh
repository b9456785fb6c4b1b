#!/bin/bash
# timeline of a graph-replayed step: bash scratch/tl.sh <tag> <bench args...>
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/db -o t -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --infer-size 0 --infer-large 0 --also-dtype none --event-steps 0 "$@" > $OUT/run.log 2>&1
python3 $R/scratch/timeline.py $OUT/db > $OUT/timeline.txt
rm -rf $OUT/db
tail -1 $OUT/run.log | cut -c1-300
