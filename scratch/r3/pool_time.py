import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spr_pick_amd import ops
for shape in ((4, 48, 4096, 4096), (4, 48, 2048, 2048), (256, 48, 64, 64)):
    x = torch.randn(shape, device="cuda")
    with torch.no_grad():
        y = ops.shift_maxpool2(x, 1); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = ops.shift_maxpool2(x, 1)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = x.numel() * 4 * 1.25 / 1e9
    print(shape, "%.3f ms  %.2f TB/s" % (ms, gb / ms))
