import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from spr_pick_amd import _lib, ops
L = _lib.lib(); d = torch.device("cuda:0")
PAD = (2, 0, 1, 1) if os.environ.get('SHIFT') else (1, 1, 1, 1)
N, C, Co, H, W = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 48, int(sys.argv[2]) if len(sys.argv) > 2 else 48, 64, 64
def run(x, gy):
    w = torch.zeros(Co, C, 3, 3, device=d)
    g = ops.make_geom(x, None, w, False, 1, 1, PAD, dtype=_lib.DTYPES["bf16!"])
    gw = torch.empty_like(w)
    nb = L.sprk_conv2d_bwd_weight_ws_bytes(ctypes.byref(g)); ws = torch.empty(nb, dtype=torch.uint8, device=d)
    n0 = L.sprk_wgrad16_launch_count()
    _lib.check(L.sprk_conv2d_bwd_weight(ops._p(x), None, ops._p(gy), ops._p(gw), ctypes.byref(g), ops._p(ws), nb, ops._stream(x)), "bw")
    torch.cuda.synchronize()
    assert L.sprk_wgrad16_launch_count() == n0 + 1
    ref = torch.nn.grad.conv2d_weight(F.pad(x.cpu().bfloat16().double(), (PAD[2], PAD[3], PAD[0], PAD[1])), w.shape, gy.cpu().bfloat16().double())
    return gw.cpu().double(), ref
torch.manual_seed(0)
for name, x, gy in (("rand", torch.randn(N, C, H, W, device=d), torch.randn(N, Co, H, W, device=d)), ("ones", torch.ones(N, C, H, W, device=d), torch.ones(N, Co, H, W, device=d)),
                    ("gy=co", torch.ones(N, C, H, W, device=d), torch.arange(Co, device=d).float().view(1, Co, 1, 1).expand(N, Co, H, W).contiguous()),
                    ("x=ci", torch.arange(C, device=d).float().view(1, C, 1, 1).expand(N, C, H, W).contiguous(), torch.ones(N, Co, H, W, device=d)),
                    ("x=col", torch.arange(W, device=d).float().view(1, 1, 1, W).expand(N, C, H, W).contiguous(), torch.ones(N, Co, H, W, device=d)),
                    ("gy=col", torch.ones(N, C, H, W, device=d), torch.arange(W, device=d).float().view(1, 1, 1, W).expand(N, Co, H, W).contiguous()),
                    ("gy=row", torch.ones(N, C, H, W, device=d), torch.arange(H, device=d).float().view(1, 1, H, 1).expand(N, Co, H, W).contiguous())):
    got, ref = run(x, gy)
    err = (got - ref).abs().max().item()
    print("%-7s max err %.4g (scale %.4g)  got[0,0]=%s ref[0,0]=%s  got[5,7]=%s" % (name, err, ref.abs().max().item(), got[0, 0].flatten().tolist(), ref[0, 0].flatten().tolist(), got[5, 7].flatten().tolist()[:3]))
# which columns are lost?  channel ci carries a delta at column ci % 64
if len(sys.argv) > 3:
    C, Co = 96, 48
    x = torch.zeros(N, C, H, W, device=d)
    for ci in range(C):
        x[:, ci, :, ci % 64] = 1.0
    got, ref = run(x, torch.ones(N, Co, H, W, device=d))
    print("x columns lost (center tap):", [ci for ci in range(64) if abs(got[0, ci, 1, 1] - ref[0, ci, 1, 1]) > 1], " partially:", [(ci, got[0, ci, 1, 1].item(), ref[0, ci, 1, 1].item()) for ci in range(64) if abs(got[0, ci, 1, 1] - ref[0, ci, 1, 1]) > 1][:6])
    gy = torch.zeros(N, 96, H, W, device=d)
    for co in range(96):
        gy[:, co, :, co % 64] = 1.0
    Co = 96; C = 48
    got, ref = run(torch.ones(N, C, H, W, device=d), gy)
    print("gy columns lost (center tap):", [(co, got[co, 0, 1, 1].item(), ref[co, 0, 1, 1].item()) for co in range(64) if abs(got[co, 0, 1, 1] - ref[co, 0, 1, 1]) > 1][:8])
