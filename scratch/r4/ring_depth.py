"""How far ahead of the device does the trainer's host thread run?  Patches feed.PinnedRing.upload to log, per call, the
number of ring slots whose copy has not completed yet and the time spent waiting.  python scratch/r4/ring_depth.py <dtype> [slots]"""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SPRK_CONV_DTYPE"] = sys.argv[1] if len(sys.argv) > 1 else "bf16"
import torch
from spr_pick_amd import cli, feed, synthetic
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 8
orig_init = feed.PinnedRing.__init__
def init(self, shape, dtype, device, slots_=slots):
    orig_init(self, shape, dtype, device, slots_)
feed.PinnedRing.__init__ = init
log = collections.defaultdict(list)
orig = feed.PinnedRing.upload
def upload(self, host_tensor, out=None):
    pending = sum(1 for e in self._events if e is not None and not e.query())
    t0 = time.perf_counter()
    r = orig(self, host_tensor, out)
    log[id(self)].append((pending, time.perf_counter() - t0, t0))
    return r
feed.PinnedRing.upload = upload
ds = synthetic.write_dataset("/tmp/lp_set", 8)
argv = ("train start -a ssdn -n gaussian --noise_value var -t %s -l %s -ap 0.75 -tau 0.01 -iter 19200 --train_batch_size 32 --nms 18 "
        "--bb 24 --runs_dir /tmp/lp_runs --print_interval 6400 --checkpoint_interval 19200 --eval_interval 19200" % (ds["images"], ds["labels"])).split()
tr = cli.start(argv)
print(tr.timing.get("loop_images"), tr.timing.get("loop_s"), "-> %.1f patches/s" % (tr.timing["loop_images"] / tr.timing["loop_s"]))
for k, v in log.items():
    v = v[len(v) // 3:]
    pend = collections.Counter(p for p, _, _ in v)
    waits = sorted(w for _, w, _ in v)
    per = (v[-1][2] - v[0][2]) / (len(v) - 1)
    print("ring %x (%d slots): calls %d, host period %.2f ms; pending copies at entry %s; upload() time median %.3f ms, p90 %.3f ms, max %.3f ms" % (
        k & 0xffff, slots, len(v), per * 1e3, dict(sorted(pend.items())), waits[len(waits) // 2] * 1e3, waits[int(len(waits) * 0.9)] * 1e3, waits[-1] * 1e3))
