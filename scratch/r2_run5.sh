#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r2_gpu_tests5.log 2>&1; tail -6 $O/r2_gpu_tests5.log | cut -c1-400
grep -h "16-bit launches" $O/r2_gpu_tests5.log
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -s -k "16bit" 2>&1 | grep "16-bit launches" | cut -c1-600
timeout -k 10 600 python bench.py --cpu-seconds 6 > $O/r2_bench5.log 2>&1; tail -c 1500 $O/r2_bench5.log
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 32 16; do
  rocprofv3 --kernel-trace --stats -d $R/$O/trace_b$B -o t -- python3 $R/bench.py --steps 10 --warmup 2 --batch $B --graph off --no-cpu-baseline --infer-size 0 --infer-large 0 --also-dtype none --event-steps 0 > $R/$O/trace_b$B.log 2>&1
  python3 $R/scratch/kstats.py $(ls $R/$O/trace_b$B/*/*.db $R/$O/trace_b$B/*.db 2>/dev/null | head -1) 17 40 > $R/$O/kstats_b$B.txt 2>&1
  rm -rf $R/$O/trace_b$B
done
head -3 $R/$O/kstats_b32.txt $R/$O/kstats_b16.txt
