"""Whole-micrograph inference A/B on one box: python scratch/r3/infer_ab.py [size] -- network ms with the fused head on/off."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from spr_pick_amd import Denoiser, DetectionDataset, networks, synthetic
from spr_pick_amd.params import PipelineOutput as P
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
torch.manual_seed(0)
den = Denoiser(bench.make_cfg(), device=dev, mode="joint")
den.eval(); den.fill()
img = torch.from_numpy(synthetic.micrograph(7, size=size)[0].astype(np.float32) / 255.0).to(dev)[None, None]
eps = torch.randn(img.shape, device=dev)
batch = DetectionDataset.make_batch(img, torch.zeros(1, 1))
ref = None
for dtype in ("f32", "f16"):
    den.set_conv_dtype(dtype)
    for fused in (False, True, False, True):
        networks.FUSED_HEAD = fused
        torch.cuda.reset_peak_memory_stats()
        with torch.no_grad():
            o = den.run_pipeline(batch, train=False, eps=eps); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                o = den.run_pipeline(batch, train=False, eps=eps)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        d = o[P.DETECT]
        if ref is None: ref = d.clone()
        print("%s %d^2 fused_head=%d  %7.1f ms  %6.1f Mpix/s  peak %.1f GB  max|score - first| %.2e" % (
            dtype, size, fused, dt * 1e3, size * size / dt / 1e6, torch.cuda.max_memory_allocated() / 1e9, float((d - ref).abs().max())), flush=True)
        del o, d
        torch.cuda.empty_cache()
