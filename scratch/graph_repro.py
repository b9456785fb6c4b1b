"""Bisect the capture_end crash seen through the trainer: which pre-capture activity matters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler; faulthandler.enable()
import numpy as np, torch
from bench import make_cfg
from spr_pick_amd import Denoiser, DetectionDataset, graph_step, synthetic, nms_device
from spr_pick_amd.params import PipelineOutput as P
what = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
torch.manual_seed(0)
den = Denoiser(make_cfg(), device=dev, mode="joint"); den.train()
opt = graph_step.make_adam([p for p in den.parameters() if p.requires_grad])
mics = [synthetic.micrograph(i) for i in range(2)]
batches = synthetic.patch_batches(4, 16, mics, seed=1, device=dev)
st = graph_step.GraphedTrainStep(den, 16, 64, 0.75, 0.01)
def evalpass():
    den.eval(); den.fill()
    with torch.no_grad():
        img = torch.rand(1, 1, 128, 128, device=dev)
        o = den.run_pipeline(DetectionDataset.make_batch(img, torch.zeros(1, 1)), train=False)
        if "nms" in what:
            nms_device(o[P.DETECT][0, 0], 18, 0.02)
    den.unfill(); den.train()
if "eval0" in what: evalpass()
keep = None
for i in range(2):
    o = st(*batches[i]); opt.step()
    if "keep" in what: keep = o
    if "metric" in what:
        from spr_pick_amd.utils import MetricDict
        md = MetricDict()
        with torch.no_grad():
            md["loss"] += o[P.LOSS]; md["a"] += o[P.DETECT_LOSS].unsqueeze(0); md["m"] += o[P.MODEL_STD_DEV] * 255
del o
print("eager ok", flush=True)
if "eval" in what: evalpass()
if "sd" in what:
    sd = opt.state_dict(); s2 = den.state_dict()
if "empty" in what: torch.cuda.empty_cache()
if "lr" in what: graph_step.set_lr(opt, 5e-5)
st(*batches[2]); opt.step()
torch.cuda.synchronize(); print("capture 1 ok", flush=True)
st(*batches[3], flip_p=0.9); opt.step()
st(*batches[3], flip_p=0.1); opt.step()
torch.cuda.synchronize(); print("all ok", what, flush=True)
