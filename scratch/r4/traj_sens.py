"""How far do two CORRECT implementations of the 10-step trajectory drift apart?  oracle in fp32 vs the same oracle in fp64
(same batches, noise, flips, ramp): the drift is the budget any fp32 implementation must be granted."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from oracle import pipeline as op, weights
from spr_pick_amd import synthetic
STEPS, B, ITER = 10, 4, 44
mics = [synthetic.micrograph(i, size=512, blobs=60) for i in range(2)]
batches = synthetic.patch_batches(STEPS, B, mics, seed=11, device="cpu")
g = torch.Generator().manual_seed(5)
eps = [(torch.randn(B, 1, 64, 64, generator=g), torch.randn(B, 1, 64, 64, generator=g)) for _ in range(STEPS)]
flips = [float(v) for v in torch.rand(STEPS, generator=g)]
state = weights.make_state(weights.denoiser_shapes(), seed=0)
def run(dt, perturb=0.0):
    sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            if perturb:
                v.mul_(1 + perturb * torch.randn(v.shape, generator=torch.Generator().manual_seed(1)).to(dt))
            v.requires_grad_(True)
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=1e-4, betas=(0.9, 0.99))
    out = []
    for i in range(STEPS):
        for grp in opt.param_groups: grp["lr"] = op.trainer_lrate(i * B, ITER)
        opt.zero_grad(set_to_none=True)
        inp, tgt = batches[i]
        r = op.joint_pipeline(sd, inp.to(dt), tgt.to(dt), 0.75, 0.01, True, eps[i][0].to(dt), eps[i][1].to(dt), flips[i])
        r["LOSS"].mean().backward(); opt.step()
        out.append({k: r[k].detach().double().clone() for k in ("LOSS", "DENOISE_LOSS", "DETECT_LOSS", "AUG_LOSS", "DETECT")})
    return out, sd
a, sda = run(torch.float32)
b, sdb = run(torch.float64)
c, sdc = run(torch.float32, perturb=1e-7)
for name, x, y in (("fp32 vs fp64", a, b), ("fp32 vs fp32 with weights perturbed by 1e-7 relative", a, c)):
    print(name)
    for i in range(STEPS):
        print(i, {k: "%.2e" % float(((x[i][k] - y[i][k]).abs().max()) / (1.0 if k == "AUG_LOSS" else max(float(y[i][k].abs().max()), 1e-3))) for k in x[i]})
