"""Filled inference checksum (to compare staging variants bit for bit): python infer_check.py SIZE"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_cfg
from spr_pick_amd import Denoiser, DetectionDataset, nms_device, synthetic
from spr_pick_amd.params import PipelineOutput as P
S = int(sys.argv[1])
torch.manual_seed(0)
den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
img = torch.from_numpy(synthetic.micrograph(7, size=S)[0].astype(np.float32) / 255.0).cuda()[None, None]
den.eval(); den.fill()
with torch.no_grad():
    torch.manual_seed(1)
    oe = den.run_pipeline(DetectionDataset.make_batch(img, torch.zeros(1, 1)), train=False)
    sm = oe[P.DETECT][0, 0].contiguous()
    s, c = nms_device(sm, 18, 0.02)
torch.cuda.synchronize()
h = hashlib.sha1(sm.cpu().numpy().tobytes()).hexdigest()[:16]
h2 = hashlib.sha1(oe[P.IMG_DENOISED].cpu().numpy().tobytes()).hexdigest()[:16]
print("size %d score %s denoised %s picks %d mem %.1f GB" % (S, h, h2, len(s), torch.cuda.max_memory_allocated() / 1e9))
