"""Why does a run whose TRAINING losses look healthy write no picks?  Load its final weights, evaluate with the ORACLE on the
CPU: (a) filled eval (running BatchNorm statistics) on a crop with planted particles; (b) unfilled patches centred on
particles / background in train mode (batch statistics) and eval mode (running statistics)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from spr_pick_amd import checkpoint, synthetic
from oracle import pipeline as op, networks as on
sd_all = checkpoint.load(sys.argv[1])
sd = {k[len("models."):]: v.float().cpu() for k, v in sd_all.items() if k.startswith("models.") and torch.is_tensor(v)}
q, centres, labelled = synthetic.micrograph(0, size=1024)
img = torch.from_numpy(q.astype(np.float32) / 255.0).T.contiguous()       # tensors enter transposed: row = x
P = 64
def patch(cx, cy):      # transposed coords: tensor[row=x, col=y]
    return img[cx - 32:cx + 32, cy - 32:cy + 32][None, None]
pos = [(int(cx), int(cy)) for cy, cx in centres[:8]]
rng = np.random.default_rng(0)
neg = [(int(a), int(b)) for a, b in rng.integers(100, 900, size=(8, 2))]
x = torch.cat([patch(*p) for p in pos + neg])
eps = torch.randn(x.shape, generator=torch.Generator().manual_seed(0))
with torch.no_grad():
    for training in (True, False):
        s = {k: v.clone() for k, v in sd.items()}
        out, logit = on.joint_forward(s, op.MODEL, x, eps, False, training)
        p = op.sigmoid_clamp(logit).reshape(-1)
        print("unfilled patches, BatchNorm %s: p(particle-centred) %s  p(background) %s" % (
            "batch stats (train)" if training else "running stats (eval)", np.round(p[:8].numpy(), 3), np.round(p[8:].numpy(), 3)))
    # which BN layer's running stats disagree with the batch stats of these patches?
    s = {k: v.clone() for k, v in sd.items()}
    for k in sorted(s):
        if k.endswith("running_var") and "detector" in k:
            print("  %-70s running_mean %.3g..%.3g running_var %.3g..%.3g" % (k[:-12], float(s[k[:-3] + "mean"].min()), float(s[k[:-3] + "mean"].max()), float(s[k].min()), float(s[k].max())))
    crop = img[:256, :256][None, None]
    e2 = torch.randn(crop.shape, generator=torch.Generator().manual_seed(1))
    r = op.joint_pipeline({k: v.clone() for k, v in sd.items()}, crop, None, 0, 0, False, e2)
    sc = r["DETECT"][0, 0]
    print("filled eval on a 256x256 crop: score min %.4f max %.4f mean %.4f" % (float(sc.min()), float(sc.max()), float(sc.mean())))
    inside = [(int(cx), int(cy)) for cy, cx in centres if 40 < cx < 216 and 40 < cy < 216]
    print("  score at planted centres:", [round(float(sc[cx, cy]), 3) for cx, cy in inside])
    # the U-Net's output level: 64x64 patch context (training) vs 256x256 crop context (filled eval), same pixels
    cx, cy = inside[0]
    pt = img[cx - 32:cx + 32, cy - 32:cy + 32][None, None]
    o_p = on.unet_blindspot(sd, op.MODEL + "denoise_branch.", pt)
    o_c = on.unet_blindspot(sd, op.MODEL + "denoise_branch.", crop)
    mu_p, mu_c = o_p[0, 0], o_c[0, 0, cx - 32:cx + 32, cy - 32:cy + 32]
    print("mu on the 64x64 patch: mean %.4f std %.4f | same pixels inside the 256x256 crop: mean %.4f std %.4f | mean |diff| %.4f" % (
        float(mu_p.mean()), float(mu_p.std()), float(mu_c.mean()), float(mu_c.std()), float((mu_p - mu_c).abs().mean())))
    print("A^2 patch mean %.5f crop mean %.5f ; detector.m running mean %.4f std %.4f" % (float((o_p[0,1]**2).mean()), float((o_c[0,1,cx-32:cx+32,cy-32:cy+32]**2).mean()),
          float(sd[op.MODEL+"detector.m.running_mean"]), float(sd[op.MODEL+"detector.m.running_var"])**0.5))
    # feed the detector (filled) with the PATCH-context z placed in the crop: does it detect then?
