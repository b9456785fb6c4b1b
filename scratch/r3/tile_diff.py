"""Where do tiled and whole-image evaluation stop being bit-identical?  (receptive field 346 px < halo 448)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import weights
from spr_pick_amd import Denoiser, synthetic, ops
sys.path.insert(0, "tests")
from test_gpu_pipeline import make_cfg

den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
den.load_state_dict({"models." + k: v for k, v in weights.make_state(weights.denoiser_shapes(), seed=0).items()}, strict=False)
den.eval(); den.fill()
S = 2048
img = torch.from_numpy(synthetic.micrograph(9, size=S)[0].astype(np.float32) / 255.0).cuda()[None, None]
net = den.models["denoiser_model"].denoise_branch
sig = den.models["sigma_estimation_model"]
det = den.models["denoiser_model"].detector
W0, WS = 512, 1024 + 2 * 448      # window origin (multiple of 32) and size
y0 = x0 = W0 - 0
win = img[:, :, W0 - 448 + 448 - 448: , :]  # placeholder
wy = wx = 64
win = img[:, :, wy:wy + WS, wx:wx + WS].contiguous()
iy = slice(448, 448 + 1024)

def cmp(name, a, b):
    a = a[..., wy + 448: wy + 448 + 1024, wx + 448: wx + 448 + 1024]
    b = b[..., iy, iy]
    d = (a - b).abs().max().item()
    print("%-28s max|d| %.3e  of %.3e  %s" % (name, d, a.abs().max().item(), "IDENTICAL" if torch.equal(a, b) else ""))

with torch.no_grad():
    cmp("unet out_stats", net(img)[0], net(win)[0])
    cmp("sigma net", sig(img), sig(win))
    z = torch.randn(1, 1, S, S, device="cuda")
    cmp("detector(filled)", det(z), det(z[:, :, wy:wy + WS, wx:wx + WS].contiguous()))
    # layer by layer through the U-Net; compare the un-rotated image 0 only (interior block, scaled)
    def cmps(name, a, b, sc):
        lo, n = (wy + 448) // sc, 1024 // sc
        aa = a[:1, :, lo:lo + n, lo:lo + n]; bb = b[:1, :, 448 // sc: 448 // sc + n, 448 // sc: 448 // sc + n]
        print("%-28s max|d| %.3e of %.3e %s" % (name, (aa - bb).abs().max().item(), aa.abs().max().item(),
                                               "IDENTICAL" if torch.equal(aa, bb) else ""))
    def run(x):
        outs = []
        e = [net.encode_block_1, net.encode_block_2, net.encode_block_3, net.encode_block_4, net.encode_block_5, net.encode_block_6]
        t = e[0][2](e[0][0](x)); outs.append(("enc1 convs", t, 1))
        p1 = net._run_pool(e[0], t); outs.append(("pool1", p1, 2))
        t = e[1][0](p1); outs.append(("enc2", t, 2)); p2 = net._run_pool(e[1], t)
        t = e[2][0](p2); outs.append(("enc3", t, 4)); p3 = net._run_pool(e[2], t)
        t = e[3][0](p3); outs.append(("enc4", t, 8)); p4 = net._run_pool(e[3], t)
        t = e[4][0](p4); outs.append(("enc5", t, 16)); p5 = net._run_pool(e[4], t); outs.append(("pool5", p5, 32))
        t = e[5][0](p5, up_out=True); outs.append(("enc6 up_out", t, 16))
        sc = 16
        for nm, blk, skip, up in (("dec5", net.decode_block_5, p4, True), ("dec4", net.decode_block_4, p3, True),
                                  ("dec3", net.decode_block_3, p2, True), ("dec2", net.decode_block_2, p1, True),
                                  ("dec1", net.decode_block_1, x, False)):
            t = blk[0](t, skip=skip); outs.append((nm + ".0", t, sc))
            t = blk[2](t, up_out=up); sc = sc // 2 if up else sc; outs.append((nm + ".2", t, sc))
        return outs
    x = ops.rot4_stack(img); xw = ops.rot4_stack(win)
    for (n1, a, sc), (_, b, _) in zip(run(x), run(xw)):
        cmps(n1, a, b, sc)
