#!/bin/bash
# GPU busy fraction of the trainer's loop (CLI) under a kernel trace: bash scratch/r4/loop_prof.sh <dtype>
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/loop_$1
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from spr_pick_amd import synthetic
synthetic.write_dataset("/tmp/lp_set", 8)
PY
PYTHONPATH=$R SPRK_CONV_DTYPE=$1 rocprofv3 --kernel-trace -d $OUT/db -o t -- python3 -m spr_pick_amd train start -a ssdn -n gaussian --noise_value var -t /tmp/lp_set/images.txt -l /tmp/lp_set/labels.txt -ap 0.75 -tau 0.01 -iter 19200 --train_batch_size 32 --nms 18 --bb 24 --runs_dir /tmp/lp_runs_$1 --print_interval 6400 --checkpoint_interval 19200 --eval_interval 19200 > $OUT/run.log 2>&1
python3 - <<PY
import sqlite3, os, collections
d="$OUT/db"
f=[os.path.join(r,x) for r,_,fs in os.walk(d) for x in fs if x.endswith(".db")][0]
con=sqlite3.connect(f)
rows=con.execute("select start, end, name from kernels order by start").fetchall()
n=len(rows); a=rows[n//3:9*n//10]
busy=sum(e-s for s,e,_ in a); wall=a[-1][1]-a[0][0]
print("dispatches %d (window %d): busy %.1f ms of wall %.1f ms = %.3f" % (n, len(a), busy/1e6, wall/1e6, busy/wall))
gaps=collections.Counter()
for (s0,e0,n0),(s1,e1,n1) in zip(a, a[1:]):
    g=s1-e0
    if g>20000: gaps[n1[:60]]+=g
pairs=collections.Counter(); cnt=collections.Counter()
for (s0,e0,n0),(s1,e1,n1) in zip(a, a[1:]):
    g=s1-e0
    if g>20000:
        k=(n0.replace("void ","").replace("(anonymous namespace)::","")[:50], n1.replace("void ","").replace("(anonymous namespace)::","")[:50]); pairs[k]+=g; cnt[k]+=1
print("idle gaps > 20 us by (kernel before -> kernel after): total ms, count, mean us")
for k,v in pairs.most_common(10): print("  %8.2f %6d %7.1f   %s  ->  %s" % (v/1e6, cnt[k], v/1e3/cnt[k], k[0], k[1]))
print("largest idle gaps BEFORE kernel (ms total):")
for k,v in gaps.most_common(8): print("  %8.2f  %s" % (v/1e6, k))
agg=collections.Counter()
for s,e,nm in a: agg[nm[:70]]+=e-s
steps=sum(1 for _,_,nm in a if "adam_multi" in nm)
print("optimiser steps in window:", steps, " wall per step %.3f ms" % (wall/1e6/max(steps,1)))
for k,v in agg.most_common(12): print("  %7.3f ms/step  %s" % (v/1e6/max(steps,1), k))
PY
rm -rf $OUT/db /tmp/lp_set /tmp/lp_runs_$1
grep "training loop" $OUT/run.log | cut -c1-200
