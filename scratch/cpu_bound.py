"""Is the training step host-bound?  Enqueue time (no sync) vs completion time of 10 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_cfg
from spr_pick_amd import Denoiser, DetectionDataset, synthetic
from spr_pick_amd.params import PipelineOutput as P
torch.manual_seed(0)
den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
mics = [synthetic.micrograph(i) for i in range(4)]
batches = synthetic.patch_batches(4, 32, mics, device="cuda:0")
params = [p for p in den.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99))
den.train()
def step(i):
    inp, tgt = batches[i % 4]
    opt.zero_grad(set_to_none=True)
    o = den.run_pipeline(DetectionDataset.make_batch(inp, tgt), 0.75, 0.01, train=True)
    torch.mean(o[P.LOSS]).backward()
    opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step, complete %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
