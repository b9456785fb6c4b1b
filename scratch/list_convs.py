"""List the convolution geometries of one training step (forward calls) with their FLOPs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_cfg
from spr_pick_amd import Denoiser, DetectionDataset, ops, synthetic
from spr_pick_amd.params import PipelineOutput as P
rec = []
orig = ops.make_geom
def mg(*a, **k):
    g = orig(*a, **k); rec.append(g); return g
ops.make_geom = mg
torch.manual_seed(0)
den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
mics = [synthetic.micrograph(i) for i in range(2)]
inp, tgt = synthetic.patch_batches(1, 32, mics, device="cuda:0")[0]
den.train()
o = den.run_pipeline(DetectionDataset.make_batch(inp, tgt), 0.75, 0.01, train=True)
tot = 0
for g in rec:
    fl = 2.0 * g.N * g.Hout * g.Wout * g.Cout * (g.C1 + g.C2) * g.KH * g.KW
    tot += fl
    print("N%4d C%3d+%3d %3dx%-3d up%d -> Cout%3d %3dx%-3d k%d s%d d%d  M=%7d  %8.3f GF" % (g.N, g.C1, g.C2, g.Hin, g.Win, g.up1, g.Cout, g.Hout, g.Wout, g.KH, g.stride, g.dil, g.N * g.Hout * g.Wout, fl / 1e9))
print(len(rec), "convs, fwd GFLOP", tot / 1e9)
