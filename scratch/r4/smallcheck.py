"""determinism and fp64 error of the small-plane fp32 convolutions (forward / backward-data through the C ABI)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from spr_pick_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(3)
SHIFT = (2, 0, 1, 1)
for N in (64, 128):
    for name, C1, C2, H, Cout in (("48->48@16", 48, 0, 16, 48), ("48->48@8", 48, 0, 8, 48), ("48->48@4", 48, 0, 4, 48), ("48->48@2", 48, 0, 2, 48),
                                  ("96+48->96@4", 96, 48, 4, 96), ("96->96@4", 96, 0, 4, 96), ("96+48->96@8", 96, 48, 8, 96), ("96->96@8", 96, 0, 8, 96),
                                  ("96->96@16", 96, 0, 16, 96), ("32->32@29 valid", 32, 0, 29, 32)):
        x = torch.randn(N, C1, H, H, device=dev); x2 = torch.randn(N, C2, H, H, device=dev) if C2 else None
        w = torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.05
        pad = (0, 0, 0, 0) if "valid" in name else SHIFT
        g = ops.make_geom(x, x2, w, False, 1, 1, pad)
        ys = [ops.conv2d_forward(x, x2, w, g) for _ in range(3)]
        det = all(torch.equal(ys[0], y) for y in ys[1:])
        xin = x if x2 is None else torch.cat([x, x2], 1)
        ref = F.conv2d(F.pad(xin.double(), (pad[2], pad[3], pad[0], pad[1])), w.double())
        err = float((ys[0].double() - ref).abs().max() / ref.abs().max())
        print("N %3d %-18s deterministic %s  max err / max|y| %.2e" % (N, name, det, err), flush=True)
