"""Layer-by-layer: GPU blind-spot U-Net with bf16 operands vs the oracle's exact model (MODEL16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import networks as onet, weights
from spr_pick_amd import networks, ops
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = weights.make_state(weights.denoiser_shapes(), seed=0)
pfx = "denoiser_model.denoise_branch."
net = networks.DualNetwork(1, 2, blindspot=True, detect=True).cuda()
net.load_state_dict({k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)})
networks.set_conv_dtype(net, dt + "!")
g = torch.Generator().manual_seed(0)
x = torch.rand(8, 1, 64, 64, generator=g)
taps = {}
onet.MODEL16["dtype"] = torch.bfloat16 if dt == "bf16" else torch.float16
with torch.no_grad():
    ref = onet.unet_blindspot(sd, pfx, x, taps)
onet.MODEL16["dtype"] = None
with torch.no_grad():
    ref32 = onet.unet_blindspot(sd, pfx, x)
    xg = ops.rot4_stack(x.cuda())
    e1 = net.encode_block_1
    a = e1[0](xg); b = e1[2](a); p1 = net._run_pool(e1, b)
    p2 = net._run_pool(net.encode_block_2, net.encode_block_2[0](p1))
    p3 = net._run_pool(net.encode_block_3, net.encode_block_3[0](p2))
    out = net(x.cuda())[0]
def cmp(name, got, want):
    got = got.cpu().double(); want = want.double()
    print("%-8s max|diff| %.3e of scale %.3e  (rel %.2e)" % (name, float((got - want).abs().max()), float(want.abs().max()), float((got - want).abs().max() / want.abs().max())))
cmp("pool1", p1, taps["pool1"]); cmp("pool2", p2, taps["pool2"]); cmp("pool3", p3, taps["pool3"])
cmp("out", out, ref); cmp("out-vs-fp32", out, ref32)
# single layer: enc1.0 output
w, bia = sd[pfx + "encode_block_1.0.weight"], sd[pfx + "encode_block_1.0.bias"]
x0 = torch.cat([onet.rot90cw(x, r) for r in (0, 90, 180, 270)], 0)
onet.MODEL16["dtype"] = torch.bfloat16 if dt == "bf16" else torch.float16
a_ref = onet._lrelu(onet.shift_conv(x0, w, bia)); onet.MODEL16["dtype"] = None
cmp("enc1.0", a, a_ref)
w2, b2 = sd[pfx + "encode_block_1.2.weight"], sd[pfx + "encode_block_1.2.bias"]
onet.MODEL16["dtype"] = torch.bfloat16 if dt == "bf16" else torch.float16
b_ref = onet._lrelu(onet.shift_conv(a.cpu(), w2, b2)); onet.MODEL16["dtype"] = None
cmp("enc1.2(from GPU a)", b, b_ref)
