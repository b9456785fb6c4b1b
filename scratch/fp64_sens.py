import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import golden
from oracle import weights, pipeline
g = golden("joint_train_w.npz")
sd = weights.make_state(weights.denoiser_shapes(), seed=0)
sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
for k, v in sd.items():
    if v.is_floating_point() and "running" not in k: v.requires_grad_(True)
res = pipeline.joint_pipeline(sd, torch.from_numpy(g["inp"]).double(), torch.from_numpy(g["target"]).double(), 0.75, 0.01, True,
        torch.from_numpy(g["eps"]).double(), torch.from_numpy(g["eps_flip"]).double(), float(g["flip_p"]))
res["LOSS"].mean().backward()
print("LOSS fp64", res["LOSS"].detach().ravel().tolist())
worst = 0
for name, p in sd.items():
    key = "grad/" + name
    if p.grad is None or key + "/idx" not in g.files: continue
    a = p.grad.numpy().ravel()
    d = np.abs(a[g[key + "/idx"]] - g[key + "/val"]).max() / g[key + "/absmax"]
    if "denoise_branch" in name and ("weight" in name):
        print("%-60s fp64-vs-reference(fp32) probe err/absmax %.2e  norm ratio %.6f" % (name[15:], d, np.linalg.norm(a) / g[key + "/norm"]))
    if "detector.m." not in name: worst = max(worst, d)
print("worst", worst)
