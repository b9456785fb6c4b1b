"""Why is conv16_tile<6> 1.8x slower inside the step than in convbench?  variants: long runs, rotating buffers, epilogue."""
import sys, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spr_pick_amd import _lib, ops
L = _lib.lib(); d = torch.device('cuda:0')
DT = _lib.DTYPES[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
N, C, H = 128, 96, 64
def run(tag, nbuf, reps, bias, real=False):
    xs = [torch.randn(N, C, H, H, device=d) for _ in range(nbuf)]
    ys = [torch.empty(N, C, H, H, device=d) for _ in range(nbuf)]
    w = torch.randn(C, C, 3, 3, device=d) * 0.05
    b = torch.randn(C, device=d) if bias else None
    g = ops.make_geom(xs[0], None, w, False, 1, 1, (2, 0, 1, 1), dtype=DT)
    ep = _lib.ConvEpilogue(ops._p(b), None, None, None, 0, 0, 0, 1, 0)
    st = ops._stream(xs[0])
    nb = L.sprk_conv2d_fwd_ws_bytes(ctypes.byref(g)); ws = ops._ws(nb, xs[0])
    def f(i):
        _lib.check(L.sprk_conv2d_fwd(ops._p(xs[i % nbuf]), None, ops._p(w), ops._p(ys[i % nbuf]), ctypes.byref(g), ctypes.byref(ep), ops._p(ws), nb, st), "f")
    f(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): f(i)
    e1.record(); torch.cuda.synchronize()
    print("%-40s %8.1f us/call" % (tag, e0.elapsed_time(e1) / reps * 1e3), flush=True)
run("1 buffer, 20 reps", 1, 20, False)
run("1 buffer, 2000 reps", 1, 2000, False)
run("8 buffers (3.2 GB), 200 reps", 8, 200, False)
run("8 buffers, 2000 reps", 8, 2000, False)
run("8 buffers, 2000 reps, bias+act", 8, 2000, True)
run("1 buffer, 20 reps (again)", 1, 20, False)
