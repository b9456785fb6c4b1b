"""16-bit-storage kernels on the hot shapes, optionally with phases switched off (diag build):
  make -C spr_pick_amd/csrc DIAG=1 BUILD=_build_diag OUT=../libsprk_diag.so -j4
  python scratch/r4/c16bench.py            (spawns itself per SPRK_C16_DIAG value when libsprk_diag.so exists)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
DIAG_LIB = os.path.join(ROOT, "spr_pick_amd", "libsprk_diag.so")
if (len(sys.argv) == 1 or sys.argv[1] != "child") and os.path.exists(DIAG_LIB):
    cases = ((0, "everything"), (1, "no input loads"), (2, "no weight DMA"), (4, "no MFMA loop"), (8, "no stores"),
             (5, "no loads, no MFMA"), (12, "no MFMA, no stores"), (13, "loads/DMA/stores off except weights"))
    if len(sys.argv) > 1:
        cases = tuple((int(v), "") for v in sys.argv[1].split(","))
    for d, what in cases:
        env = dict(os.environ, SPRK_LIB=DIAG_LIB, SPRK_C16_DIAG=str(d))
        print("== SPRK_C16_DIAG=%d (%s)" % (d, what), flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env)
    sys.exit(0)
import torch
from spr_pick_amd import _lib, ops
L = _lib.lib()
dev = torch.device("cuda:0")
tdt = torch.bfloat16
SH = {"96->96@64 N128": (128, 96, 0, 64, 64, 96, 3, (2, 0, 1, 1)), "48->48@64 N128": (128, 48, 0, 64, 64, 48, 3, (2, 0, 1, 1)),
      "96+48->96@32 N128": (128, 96, 48, 32, 32, 96, 3, (2, 0, 1, 1)), "1x1 384->384@64 N32": (32, 384, 0, 64, 64, 384, 1, (0, 0, 0, 0))}
for name, (N, C1, C2, H, W, Cout, K, pad) in SH.items():
    x = torch.randn(N, C1, H, W, device=dev).to(tdt)
    x2 = torch.randn(N, C2, H, W, device=dev).to(tdt) if C2 else None
    w = torch.randn(Cout, C1 + C2, K, K, device=dev) * 0.05
    g = ops.make_geom(x, x2, w, False, 1, 1, pad, dtype=_lib.DT_BF16 | _lib.DT_X16 | _lib.DT_Y16)
    gw_ = ops.make_geom(x, x2, w, False, 1, 1, pad, dtype=_lib.DT_BF16 | _lib.DT_X16)
    y = torch.empty(N, Cout, H, W, device=dev, dtype=tdt); gy = torch.randn(N, Cout, H, W, device=dev).to(tdt)
    gin = torch.empty(N, C1 + C2, H, W, device=dev, dtype=tdt); gw = torch.empty_like(w)
    ep = _lib.ConvEpilogue(None, None, None, None, 0, 0, 0, 1)
    st = ops._stream(x)
    def fwd():
        nb = L.sprk_conv2d_fwd_ws_bytes(ctypes.byref(g)); ws = ops._ws(nb, x)
        _lib.check(L.sprk_conv2d_fwd(ops._p(x), ops._p(x2), ops._p(w), ops._p(y), ctypes.byref(g), ctypes.byref(ep), ops._p(ws), nb, st), "f")
    def bd():
        nb = L.sprk_conv2d_bwd_data_ws_bytes(ctypes.byref(g)); ws = ops._ws(nb, x)
        _lib.check(L.sprk_conv2d_bwd_data_masked(ops._p(gy), ops._p(w), ops._p(gin), ctypes.byref(g), None, 0, ops._p(ws), nb, st), "b")
    def bw():
        nb = L.sprk_conv2d_bwd_weight_ws_bytes(ctypes.byref(gw_)); ws = ops._ws(nb, x)
        _lib.check(L.sprk_conv2d_bwd_weight(ops._p(x), ops._p(x2), ops._p(gy), ops._p(gw), ctypes.byref(gw_), ops._p(ws), nb, st), "w")
    res = []
    for fn in (fwd, bd, bw):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.append("%s %7.1f us" % (fn.__name__, e0.elapsed_time(e1) / 20 * 1e3))
    io = 2.0 * (x.numel() + (x2.numel() if x2 is not None else 0) + y.numel()) / 1e6
    print("%-22s %s | io %.0f MB" % (name, " | ".join(res), io), flush=True)
