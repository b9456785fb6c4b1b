#!/bin/bash
# round-2 GPU session 1: graph-captured step — tests, then bench variants
cd $GRAFT_REPO_ROOT
O=gpurun_out
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --infer-size 0 --infer-large 0 > $O/r2_g_on32.log 2>&1; tail -c 600 $O/r2_g_on32.log; echo
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --infer-size 0 --infer-large 0 --batch 16 > $O/r2_g_on16.log 2>&1; tail -c 300 $O/r2_g_on16.log; echo
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --infer-size 0 --infer-large 0 --graph off > $O/r2_g_off32.log 2>&1; tail -c 300 $O/r2_g_off32.log; echo
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --infer-size 0 --infer-large 0 --graph off --batch 16 > $O/r2_g_off16.log 2>&1; tail -c 300 $O/r2_g_off16.log; echo
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r2_gpu_tests1.log 2>&1; tail -15 $O/r2_gpu_tests1.log
