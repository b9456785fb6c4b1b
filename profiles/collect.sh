#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run from the repo root through gpurun):
#   bash profiles/collect.sh r01
# runs the bench plain, under the kernel trace and under three PMC passes, then profiles/summarize.py on the
# databases; gpurun_out/prof_<tag>/summary/ holds the files to copy into profiles/.
# (PMC passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes.)
# PARTS="16" re-collects only the bf16 passes (the other summaries under profiles/ are kept).
set -e
TAG=${1:-r01}
PARTS=${PARTS:-all}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
if [ "$PARTS" = "all" ]; then
python3 $R/bench.py --cpu-seconds 12 > $OUT/bench.log 2> $OUT/bench.err
grep '^{"metric"' $OUT/bench.log > $OUT/bench.json
fi
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --infer-size 0 --infer-large 0 --also-dtype none --graph off --sustain-seconds 0 --batch16 off --full-pipeline off"
if [ "$PARTS" = "all" ]; then
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
grep '^{"metric"' $OUT/trace.log > $OUT/bench_under_rocprof.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $OUT/mfma -o m -- python3 $R/bench.py $ARGS > $OUT/mfma.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/infer -o i -- python3 $R/scratch/infer_prof.py 4096 > $OUT/infer.log 2>&1
fi
rocprofv3 --kernel-trace --stats -d $OUT/trace16 -o t -- python3 $R/bench.py $ARGS --dtype bf16 --also-dtype none > $OUT/trace16.log 2>&1
# the 16-bit kernels of the bf16 step: the same three PMC passes (round 4: train_bf16.roofline.traffic)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch16 -o f -- python3 $R/bench.py $ARGS --dtype bf16 > $OUT/fetch16.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write16 -o w -- python3 $R/bench.py $ARGS --dtype bf16 > $OUT/write16.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $OUT/mfma16 -o m -- python3 $R/bench.py $ARGS --dtype bf16 > $OUT/mfma16.log 2>&1
mkdir -p $OUT/summary
if [ "$PARTS" != "all" ]; then cp $R/profiles/${TAG}_*.json $R/profiles/${TAG}_*.csv $OUT/summary/ 2>/dev/null || true; fi
python3 $R/profiles/summarize.py $OUT $TAG $OUT/summary
rm -rf $OUT/trace $OUT/trace16 $OUT/fetch $OUT/write $OUT/mfma $OUT/infer $OUT/fetch16 $OUT/write16 $OUT/mfma16
ls $OUT/summary
