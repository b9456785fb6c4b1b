#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "512 512" "512 256" "512 128" "256 256" "256 128" "512 64"; do
  set -- $cfg
  for gb in 32 16; do
    SPRK_FWD_MINBLK_M=$1 SPRK_FWD_MINBLK_N=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --infer-size 0 --infer-large 0 --global-batch $gb --steps 60 --warmup 3 --event-steps 2 > gpurun_out/sw.log 2>&1
    tail -1 gpurun_out/sw.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('M $1 N $2 batch $gb: f32 %.1f (%.2f ms)  bf16 %.1f (%.2f ms)' % (j['value'], j['ms_per_step'], j['train_bf16']['value'], j['train_bf16']['ms_per_step']))"
  done
done
