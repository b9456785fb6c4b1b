import sys, ctypes
sys.path.insert(0, '.')
import torch, numpy as np
from spr_pick_amd import _lib, ops
L = _lib.lib()
torch.manual_seed(0)
d = torch.device('cuda:0')
def run(N, C1, C2, H, W, up1, Cout, K, pad):
    h1, w1 = (H//2, W//2) if up1 else (H, W)
    x = torch.randn(N, C1, h1, w1, device=d); x2 = torch.randn(N, C2, H, W, device=d) if C2 else None
    w = torch.randn(Cout, C1+C2, K, K, device=d) / np.sqrt((C1+C2)*K*K)
    g = ops.make_geom(x, x2, w, bool(up1), 1, 1, pad)
    gy = torch.randn(N, Cout, g.Hout, g.Wout, device=d)
    outs = []
    for naive in (1, 0):
        L.sprk_set_naive(naive)
        gin = torch.full((N, C1+C2, H, W), float('nan'), device=d)
        nb = L.sprk_conv2d_bwd_data_ws_bytes(ctypes.byref(g)); ws = torch.empty(max(nb,16), dtype=torch.uint8, device=d)
        _lib.check(L.sprk_conv2d_bwd_data(ops._p(gy), ops._p(w), ops._p(gin), ctypes.byref(g), ops._p(ws), nb, ops._stream()), "bd")
        torch.cuda.synchronize()
        outs.append(gin.cpu())
    L.sprk_set_naive(0)
    a, b = outs
    err = (a-b).abs()
    bad = err > 1e-3
    print("case", (N,C1,C2,H,W,up1,Cout,K,pad), "nan:", int(torch.isnan(b).sum()), "bad:", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()
        print(" n:", idx[:,0].unique().tolist()[:10], "c:", idx[:,1].unique().tolist()[:40])
        print(" y:", idx[:,2].unique().tolist()[:70]); print(" x:", idx[:,3].unique().tolist()[:70])
        print(" first:", idx[:12].tolist())
for N in (1,2,3,4):
    run(N, 96, 1, 64, 64, 1, 96, 3, (1,1,1,1))
run(2, 96, 1, 64, 64, 1, 96, 3, (2,0,1,1))
run(2, 97, 0, 64, 64, 0, 96, 3, (1,1,1,1))
run(2, 96, 0, 64, 64, 0, 96, 3, (1,1,1,1))
