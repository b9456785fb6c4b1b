import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_cfg
from spr_pick_amd import Denoiser, DetectionDataset, nms_device, synthetic
from spr_pick_amd.params import PipelineOutput as P
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
img = torch.from_numpy(synthetic.micrograph(7, size=S)[0].astype(np.float32) / 255.0).cuda()[None, None]
den.eval(); den.fill()
with torch.no_grad():
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        oe = den.run_pipeline(DetectionDataset.make_batch(img, torch.zeros(1, 1)), train=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        s, c = nms_device(oe[P.DETECT][0, 0], 18, 0.02)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print("size %d: pipeline %.1f ms, nms %.1f ms, picks %d, mem %.1f GB" % (S, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(s), torch.cuda.max_memory_allocated() / 1e9), flush=True)
