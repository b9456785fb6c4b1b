"""Patches/s through the real trainer loop (device feed + sampler + logging) vs bench.py's resident batches."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spr_pick_amd import cli, micrograph_io, synthetic

root = tempfile.mkdtemp()
lines, labels = ["image_name\tpath"], ["image_name\tx_coord\ty_coord"]
for k in range(4):
    q, centres, lab = synthetic.micrograph(k, size=1024)
    p = os.path.join(root, "m%d.mrc" % k)
    with open(p, "wb") as f:
        micrograph_io.write_mrc(f, q.astype(np.float32))
    lines.append("m%d\t%s" % (k, p))
    labels += ["m%d\t%d\t%d" % (k, cx, cy) for cy, cx in lab]
open(os.path.join(root, "imgs.txt"), "w").write("\n".join(lines) + "\n")
open(os.path.join(root, "lab.txt"), "w").write("\n".join(labels) + "\n")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
argv = ("train start -a ssdn -n gaussian --noise_value var -t %s -l %s -ap 0.75 -tau 0.01 -iter %d "
        "--train_batch_size 32 --print_interval 640 --checkpoint_interval 1000000 --eval_interval 1000000 --runs_dir %s"
        % (os.path.join(root, "imgs.txt"), os.path.join(root, "lab.txt"), iters, os.path.join(root, "runs"))).split()
t0 = time.time()
tr = cli.start(argv)
torch.cuda.synchronize()
print("total wall %.2f s" % (time.time() - t0))
log = open(os.path.join(tr.run_dir_path, "log.txt")).read().splitlines()
for l in log:
    if "TRAIN |" in l: print(l)
