"""Micro-benchmark of the conv kernels on the hot shapes (events on the launch stream)."""
import sys, ctypes, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from spr_pick_amd import _lib, ops
L = _lib.lib()
d = torch.device('cuda:0')
SHAPES = {
    # name: N, C1, C2, H, W, up1, Cout, K, pad
    "dec1.2 96->96@64": (128, 96, 0, 64, 64, 0, 96, 3, (2, 0, 1, 1)),
    "dec1.0 up96+1->96@64": (128, 96, 1, 64, 64, 1, 96, 3, (2, 0, 1, 1)),
    "dec2.0 up96+48->96@32": (128, 96, 48, 32, 32, 1, 96, 3, (2, 0, 1, 1)),
    "enc1.2 48->48@64": (128, 48, 0, 64, 64, 0, 48, 3, (2, 0, 1, 1)),
    "head 384->384 1x1@64": (32, 384, 0, 64, 64, 0, 384, 1, (0, 0, 0, 0)),
    "head 384->96 1x1@64": (32, 384, 0, 64, 64, 0, 96, 1, (0, 0, 0, 0)),
    "dec3.0 up96+48->96@16": (128, 96, 48, 16, 16, 1, 96, 3, (2, 0, 1, 1)),
    "net dec1.0 96+1->96@64": (128, 96, 1, 64, 64, 0, 96, 3, (2, 0, 1, 1)),
    "net dec2.0 96+48->96@32": (128, 96, 48, 32, 32, 0, 96, 3, (2, 0, 1, 1)),
    "net dec3.0 96+48->96@16": (128, 96, 48, 16, 16, 0, 96, 3, (2, 0, 1, 1)),
    "net dec3.0 96+48->96@16 N256": (256, 96, 48, 16, 16, 0, 96, 3, (2, 0, 1, 1)),
    "net dec3.2 96->96@16 N256": (256, 96, 0, 16, 16, 0, 96, 3, (2, 0, 1, 1)),
    "sig dec2.0 96+48->96@32 N64": (64, 96, 48, 32, 32, 0, 96, 3, (1, 1, 1, 1)),
    "net dec2.0 96+48->96@32 N256": (256, 96, 48, 32, 32, 0, 96, 3, (2, 0, 1, 1)),
    "net dec2.2 96->96@32 N256": (256, 96, 0, 32, 32, 0, 96, 3, (2, 0, 1, 1)),
    "sig dec1.2 96->96@64 N32": (32, 96, 0, 64, 64, 0, 96, 3, (1, 1, 1, 1)),
    "sig dec1.0 96+1->96@64 N32": (32, 96, 1, 64, 64, 0, 96, 3, (1, 1, 1, 1)),
    "net dec1.2 96->96@64 N128": (128, 96, 0, 64, 64, 0, 96, 3, (2, 0, 1, 1)),
    "det32->32@29 valid": (32, 32, 0, 29, 29, 0, 32, 3, (0, 0, 0, 0)),
    "det64->64@9 valid": (32, 64, 0, 9, 9, 0, 64, 3, (0, 0, 0, 0)),
    "enc48->48@4": (256, 48, 0, 4, 4, 0, 48, 3, (2, 0, 1, 1)),
    "enc48->48@8": (256, 48, 0, 8, 8, 0, 48, 3, (2, 0, 1, 1)),
    "inf48->48@1024": (4, 48, 0, 1024, 1024, 0, 48, 3, (2, 0, 1, 1)),
    "inf48->48@512": (4, 48, 0, 512, 512, 0, 48, 3, (2, 0, 1, 1)),
    "enc48->48@32": (128, 48, 0, 32, 32, 0, 48, 3, (2, 0, 1, 1)),
    "cin4->96@64": (128, 4, 0, 64, 64, 0, 96, 3, (2, 0, 1, 1)),
    "cin8->96@64": (128, 8, 0, 64, 64, 0, 96, 3, (2, 0, 1, 1)),
    "cin16->96@64": (128, 16, 0, 64, 64, 0, 96, 3, (2, 0, 1, 1)),
    "cin32->96@64": (128, 32, 0, 64, 64, 0, 96, 3, (2, 0, 1, 1)),
    "cin64->96@64": (128, 64, 0, 64, 64, 0, 96, 3, (2, 0, 1, 1)),
}
DT = 0
args = sys.argv[1:]
if args and args[0] in _lib.DTYPES:
    DT = _lib.DTYPES[args.pop(0)]
which = args or list(SHAPES)
reps = 20
for name in which:
    N, C1, C2, H, W, up1, Cout, K, pad = SHAPES[name]
    h1, w1 = (H // 2, W // 2) if up1 else (H, W)
    x = torch.randn(N, C1, h1, w1, device=d); x2 = torch.randn(N, C2, H, W, device=d) if C2 else None
    w = torch.randn(Cout, C1 + C2, K, K, device=d) * 0.05
    g = ops.make_geom(x, x2, w, bool(up1), 1, 1, pad, dtype=DT)
    y = torch.empty(N, Cout, g.Hout, g.Wout, device=d); gy = torch.randn_like(y)
    gin = torch.empty(N, C1 + C2, H, W, device=d); gw = torch.empty_like(w)
    flops = 2.0 * N * g.Hout * g.Wout * Cout * (C1 + C2) * K * K
    ep = _lib.ConvEpilogue(None, None, None, None, 0, 0, 0, 1)
    st = ops._stream(x)
    def fwd():
        nb = L.sprk_conv2d_fwd_ws_bytes(ctypes.byref(g)); ws = ops._ws(nb, x)
        _lib.check(L.sprk_conv2d_fwd(ops._p(x), ops._p(x2), ops._p(w), ops._p(y), ctypes.byref(g), ctypes.byref(ep), ops._p(ws), nb, st), "f")
    def bd():
        nb = L.sprk_conv2d_bwd_data_ws_bytes(ctypes.byref(g)); ws = ops._ws(nb, x)
        _lib.check(L.sprk_conv2d_bwd_data(ops._p(gy), ops._p(w), ops._p(gin), ctypes.byref(g), ops._p(ws), nb, st), "b")
    def bw():
        nb = L.sprk_conv2d_bwd_weight_ws_bytes(ctypes.byref(g)); ws = ops._ws(nb, x)
        _lib.check(L.sprk_conv2d_bwd_weight(ops._p(x), ops._p(x2), ops._p(gy), ops._p(gw), ctypes.byref(g), ops._p(ws), nb, st), "w")
    res = []
    for fn in (fwd, bd, bw):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append("%s %7.1f us %6.1f TF" % (fn.__name__, ms * 1e3, flops / ms / 1e9))
    nbytes = 4.0 * (x.numel() + (x2.numel() if x2 is not None else 0) + y.numel())
    res.append("io %.0f MB" % (nbytes / 1e6))
    print("%-26s %6.1f GFLOP | %s" % (name, flops / 1e9, " | ".join(res)), flush=True)
