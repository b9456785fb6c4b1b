import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import golden
from oracle import weights, pipeline
from test_gpu_pipeline import make_cfg
from spr_pick_amd import Denoiser, DetectionDataset, _lib
from spr_pick_amd.params import PipelineOutput as P
g = golden("joint_train_w.npz")
sd0 = weights.make_state(weights.denoiser_shapes(), seed=0)
den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
den.load_state_dict({"models." + k: v for k, v in sd0.items()}, strict=False)
den.train()
o = den.run_pipeline(DetectionDataset.make_batch(torch.from_numpy(g["inp"]), torch.from_numpy(g["target"])), 0.75, 0.01, train=True,
    eps=torch.from_numpy(g["eps"]).cuda(), eps_flip=torch.from_numpy(g["eps_flip"]).cuda(), flip_p=float(g["flip_p"]))
torch.mean(o[P.LOSS]).backward()
gp = {n: p.grad.detach().cpu().double() for n, p in den.models.named_parameters() if p.grad is not None}
sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
for k, v in sd.items():
    if v.is_floating_point() and "running" not in k: v.requires_grad_(True)
torch.set_num_threads(16)
res = pipeline.joint_pipeline(sd, torch.from_numpy(g["inp"]).double(), torch.from_numpy(g["target"]).double(), 0.75, 0.01, True,
        torch.from_numpy(g["eps"]).double(), torch.from_numpy(g["eps_flip"]).double(), float(g["flip_p"]))
res["LOSS"].mean().backward()
for name in ("denoiser_model.denoise_branch.output_block.2.bias", "denoiser_model.denoise_branch.output_block.0.bias", "denoiser_model.denoise_branch.decode_block_1.2.bias"):
    a, b = gp[name], sd[name].grad
    e = (a - b).abs()
    print(name, "absmax", float(b.abs().max()), "max err", float(e.max()), "n(err>1e-5*absmax)", int((e > 1e-5 * b.abs().max()).sum()), "of", e.numel())
    top = torch.topk(e, 5)
    print("   top errs", [(int(i), float(v), float(b[i])) for v, i in zip(top.values, top.indices)])
