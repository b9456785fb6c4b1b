"""Trainer-path variant of graph_repro.py: python scratch/graph_repro2.py <val 0|1> <eval_interval> <ckpt_interval> <print_interval>"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import faulthandler; faulthandler.enable()
from test_gpu_trainer import _write_set
from spr_pick_amd import cli
val, ei, ci, pi = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
tmp = tempfile.mkdtemp()
imgs, lab = _write_set(tmp)
argv = ("train start -a ssdn -n gaussian --noise_value var -t %s -l %s %s -ap 0.75 -tau 0.01 -iter 96 --train_batch_size 16 "
        "--eval_interval %s --print_interval %s --checkpoint_interval %s --nms 18 --bb 24 --runs_dir %s"
        % (imgs, lab, ("-v %s -vl %s" % (imgs, lab)) if val else "", ei, pi, ci, os.path.join(tmp, "runs"))).split()
cli.start(argv)
print("all ok", sys.argv[1:], flush=True)
