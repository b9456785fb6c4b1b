"""fp32 convolutions of the U-Net levels <= 16x16 (and the detector's patch layers): forward, backward-data,
backward-weight launch times at batch 16 and 32 per GPU (x4 rotations).  python scratch/r4/smallbench.py [N ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spr_pick_amd import _lib, ops
L = _lib.lib()
dev = torch.device("cuda:0")
SHIFT = (2, 0, 1, 1)
# name: (C1, up1, C2, H (output level), Cout, K, pad, stride, dil)
SH = [("48->48@16", 48, 0, 0, 16, 48, 3, SHIFT, 1, 1), ("48->48@8", 48, 0, 0, 8, 48, 3, SHIFT, 1, 1),
      ("48->48@4", 48, 0, 0, 4, 48, 3, SHIFT, 1, 1), ("48->48@2", 48, 0, 0, 2, 48, 3, SHIFT, 1, 1),
      ("48^+48->96@4", 48, 1, 48, 4, 96, 3, SHIFT, 1, 1), ("96->96@4", 96, 0, 0, 4, 96, 3, SHIFT, 1, 1),
      ("96^+48->96@8", 96, 1, 48, 8, 96, 3, SHIFT, 1, 1), ("96->96@8", 96, 0, 0, 8, 96, 3, SHIFT, 1, 1),
      ("96^+48->96@16", 96, 1, 48, 16, 96, 3, SHIFT, 1, 1), ("96->96@16", 96, 0, 0, 16, 96, 3, SHIFT, 1, 1),
      ("det 32->32 d2 @25", 32, 0, 0, 29, 32, 3, (0, 0, 0, 0), 1, 1), ("det 64->64 d2 @6", 64, 0, 0, 8, 64, 3, (0, 0, 0, 0), 1, 2)]
Ns = [int(v) for v in sys.argv[1:]] or [64, 128]
for N in Ns:
    print("== N = %d images" % N)
    tot = [0.0, 0.0, 0.0]
    for name, C1, up1, C2, H, Cout, K, pad, stride, dil in SH:
        Nn = N // 4 if name.startswith("det") else N
        hin = H // 2 if up1 else H
        x = torch.randn(Nn, C1, hin, hin, device=dev)
        x2 = torch.randn(Nn, C2, H, H, device=dev) if C2 else None
        w = torch.randn(Cout, C1 + C2, K, K, device=dev) * 0.05
        g = ops.make_geom(x, x2, w, bool(up1), stride, dil, pad)
        y = torch.empty(Nn, Cout, g.Hout, g.Wout, device=dev); gy = torch.randn_like(y)
        gin = torch.empty(Nn, C1 + C2, H, H, device=dev); gw = torch.empty_like(w)
        ep = _lib.ConvEpilogue(None, None, None, None, 0, 0, 0, 1)
        st = ops._stream(x)
        gd = ops.make_geom(gin, None, w, False, stride, dil, pad)   # backward-data: gradient of the concatenated input
        def fwd():
            nb = L.sprk_conv2d_fwd_ws_bytes(ctypes.byref(g)); ws = ops._ws(nb, x)
            _lib.check(L.sprk_conv2d_fwd(ops._p(x), ops._p(x2), ops._p(w), ops._p(y), ctypes.byref(g), ctypes.byref(ep), ops._p(ws), nb, st), "f")
        def bd():
            nb = L.sprk_conv2d_bwd_data_ws_bytes(ctypes.byref(gd)); ws = ops._ws(nb, x)
            _lib.check(L.sprk_conv2d_bwd_data_masked(ops._p(gy), ops._p(w), ops._p(gin), ctypes.byref(gd), None, 0, ops._p(ws), nb, st), "b")
        def bw():
            nb = L.sprk_conv2d_bwd_weight_ws_bytes(ctypes.byref(g)); ws = ops._ws(nb, x)
            _lib.check(L.sprk_conv2d_bwd_weight(ops._p(x), ops._p(x2), ops._p(gy), ops._p(gw), ctypes.byref(g), ops._p(ws), nb, st), "w")
        res = []
        for i, fn in enumerate((fwd, bd, bw)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 50 * 1e3
            tot[i] += t
            res.append("%s %6.1f us" % (fn.__name__, t))
        if hasattr(L, "sprk_diag_conv_clock") or os.environ.get("SPRK_LIB"):
            buf = (ctypes.c_longlong * 40)()
            for fn in (fwd, bd):
                fn(); torch.cuda.synchronize()
                ctypes.CDLL(os.environ["SPRK_LIB"]).sprk_diag_conv_clock(buf)
                res.append("[" + " ".join("%d" % ((buf[i] - buf[0]) * 10) for i in range(1, 8)) + " ns; chunks (landed, next issued, done): " + " ".join("%d" % ((buf[i] - buf[0]) * 10) for i in range(8, 8 + 3 * min(8, -(-(C1 + C2) // 32)))) + "]")
        fl = 2.0 * Nn * g.Hout * g.Wout * Cout * (C1 + C2) * K * K / 1e6
        print("%-20s %s | %.0f MFLOP" % (name, " | ".join(res), fl), flush=True)
    print("%-20s fwd %6.1f us | bd %6.1f us | bw %6.1f us   (launches include weight preparation)" % ("sum", *tot))
