#!/bin/bash
# final validation of the round: GPU suite, smoke, 2-rank gloo rehearsal of the bench on one GPU, profile collection
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r2_gpu_tests_final.log 2>&1; tail -3 $O/r2_gpu_tests_final.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SPRK_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 30 --warmup 3 --no-cpu-baseline --infer-size 0 --infer-large 0 > $O/r2_bench_2rank_gloo.log 2>&1; grep '^{"metric"' $O/r2_bench_2rank_gloo.log | cut -c1-220
bash profiles/collect.sh r02 2>&1 | tail -3
ls gpurun_out/prof_r02/summary/
