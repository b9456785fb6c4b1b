#!/bin/bash
# trainer loop rate through the CLI (short run: ~1 s of warm-up and graph capture is part of it; scratch/r4/ring_depth.py
# gives the steady-state host period): bash scratch/r4/ring_modes.sh <dtype>
R=$GRAFT_REPO_ROOT
cd /tmp
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from spr_pick_amd import synthetic
synthetic.write_dataset("/tmp/lp_set", 8)
PY
for m in run; do
  PYTHONPATH=$R SPRK_RING_MODE=$m SPRK_CONV_DTYPE=$1 python3 -m spr_pick_amd train start -a ssdn -n gaussian --noise_value var -t /tmp/lp_set/images.txt -l /tmp/lp_set/labels.txt -ap 0.75 -tau 0.01 -iter 25600 --train_batch_size 32 --nms 18 --bb 24 --runs_dir /tmp/lp_runs_$m --print_interval 6400 --checkpoint_interval 25600 --eval_interval 25600 2>&1 | grep "training loop" | sed "s/^/$m: /" | cut -c1-160
done
