"""Per-round NMS cost: python scratch/r3/nms_rounds.py [size] (needs /tmp/score.npy from nms_time.py or recomputes)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from spr_pick_amd import Denoiser, DetectionDataset, _lib, algorithms, synthetic
from spr_pick_amd.params import PipelineOutput as P
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
den = Denoiser(bench.make_cfg(), device=dev, mode="joint"); den.eval(); den.fill()
img = torch.from_numpy(synthetic.micrograph(7, size=size)[0].astype(np.float32) / 255.0).to(dev)[None, None]
with torch.no_grad():
    o = den.run_pipeline(DetectionDataset.make_batch(img, torch.zeros(1, 1)), train=False, eps=torch.randn(img.shape, device=dev))
score = o[P.DETECT][0, 0].contiguous()
L = _lib.lib()
H, W = score.shape
cap = algorithms._max_picks(H, W, 18)
out_s = torch.empty(cap, device=dev); out_xy = torch.empty((cap, 2), dtype=torch.int32, device=dev)
cnt = torch.zeros(2, dtype=torch.int32, device=dev)
ws = torch.empty(L.sprk_nms2d_ws_bytes(H, W, cap), dtype=torch.uint8, device=dev)
for rep in range(2):
    resume = 0
    for rnd in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        torch.ops.sprk.nms2d(score, 18, 0.02, out_s, out_xy, cnt, 1, resume, ws)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n, und = cnt.tolist()
        if rep: print("round %2d: %.3f ms (incl. count + sort + emit), picks %d, undecided %d" % (rnd + 1, dt * 1e3, n, und))
        resume = 1
        if und == 0: break
# cost of the tail (count + fill + sort + emit) alone: a call with 0 rounds
torch.cuda.synchronize(); t0 = time.perf_counter()
torch.ops.sprk.nms2d(score, 18, 0.02, out_s, out_xy, cnt, 0, 1, ws)
torch.cuda.synchronize(); print("0 rounds (count + sort + emit): %.3f ms" % ((time.perf_counter() - t0) * 1e3))
