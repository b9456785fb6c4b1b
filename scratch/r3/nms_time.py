"""NMS timing on the network's own score map: python scratch/r3/nms_time.py [size]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from spr_pick_amd import Denoiser, DetectionDataset, algorithms, nms_device, synthetic
from spr_pick_amd.params import PipelineOutput as P
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
den = Denoiser(bench.make_cfg(), device=dev, mode="joint"); den.eval(); den.fill()
img = torch.from_numpy(synthetic.micrograph(7, size=size)[0].astype(np.float32) / 255.0).to(dev)[None, None]
with torch.no_grad():
    o = den.run_pipeline(DetectionDataset.make_batch(img, torch.zeros(1, 1)), train=False, eps=torch.randn(img.shape, device=dev))
score = o[P.DETECT][0, 0].contiguous()
print("candidates above 0.02: %.1f %%" % (100 * float((score > 0.02).float().mean())))
for rpc in (algorithms.ROUNDS_PER_CALL, 1):
    algorithms.ROUNDS_PER_CALL = rpc
    s, c = nms_device(score, 18, 0.02); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        s, c = nms_device(score, 18, 0.02)
    torch.cuda.synchronize()
    print("%d^2 rounds/call %d: %.2f ms, %d picks" % (size, rpc, (time.perf_counter() - t0) / 5 * 1e3, len(s)))
np.save("/tmp/score.npy", score.cpu().numpy())
