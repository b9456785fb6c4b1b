"""host-side profile of the CLI trainer loop: python scratch/r4/loop_cprofile.py <dtype>"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SPRK_CONV_DTYPE"] = sys.argv[1] if len(sys.argv) > 1 else "f32"
from spr_pick_amd import cli, synthetic
ds = synthetic.write_dataset("/tmp/lp_set", 8)
argv = ("train start -a ssdn -n gaussian --noise_value var -t %s -l %s -ap 0.75 -tau 0.01 -iter 25600 --train_batch_size 32 --nms 18 "
        "--bb 24 --runs_dir /tmp/lp_runs --print_interval 6400 --checkpoint_interval 25600 --eval_interval 25600" % (ds["images"], ds["labels"])).split()
pr = cProfile.Profile()
pr.enable()
tr = cli.start(argv)
pr.disable()
print(tr.timing)
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(45)
