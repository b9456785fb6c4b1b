#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r2_gpu_tests_final.log 2>&1; tail -4 $O/r2_gpu_tests_final.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash profiles/collect.sh r02 2>&1 | tail -3
ls gpurun_out/prof_r02/summary/
