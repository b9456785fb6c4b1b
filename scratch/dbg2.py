import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import golden
from oracle import weights
from test_gpu_pipeline import make_cfg
from spr_pick_amd import Denoiser, DetectionDataset, _lib
from spr_pick_amd.params import PipelineOutput as P
naive = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_lib.lib().sprk_set_naive(naive)
sd = weights.make_state(weights.denoiser_shapes(), seed=0)
g = golden("joint_train_w.npz")
den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
den.load_state_dict({"models." + k: v for k, v in sd.items()}, strict=False)
den.train()
o = den.run_pipeline(DetectionDataset.make_batch(torch.from_numpy(g["inp"]), torch.from_numpy(g["target"])), 0.75, 0.01, train=True,
    eps=torch.from_numpy(g["eps"]).cuda(), eps_flip=torch.from_numpy(g["eps_flip"]).cuda(), flip_p=float(g["flip_p"]))
torch.mean(o[P.LOSS]).backward()
print("naive", naive, "LOSS", o[P.LOSS].detach().cpu().ravel().tolist(), g["LOSS"].ravel().tolist())
for name, p in den.models.named_parameters():
    if p.grad is None: continue
    key = "grad/" + name
    a = p.grad.detach().cpu().numpy().astype(np.float64).ravel()
    d = np.abs(a[g[key + "/idx"]] - g[key + "/val"]).max()
    print("%-70s absmax %.3e  probe err/absmax %.2e  norm ratio %.6f" % (name, g[key+"/absmax"], d / g[key+"/absmax"], np.linalg.norm(a) / g[key+"/norm"]))
