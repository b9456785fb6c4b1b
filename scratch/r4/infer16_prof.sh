#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/inf16
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/inf16.py <<PY
import sys, time; sys.path.insert(0, "$R")
import numpy as np, torch
from bench import make_cfg
from spr_pick_amd import Denoiser, DetectionDataset, nms_device, synthetic
from spr_pick_amd.params import PipelineOutput as P
S = 4096
torch.manual_seed(0)
den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
den.set_conv_dtype("f16")
img = torch.from_numpy(synthetic.micrograph(7, size=S)[0].astype(np.float32) / 255.0).cuda()[None, None]
den.eval(); den.fill()
with torch.no_grad():
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        oe = den.run_pipeline(DetectionDataset.make_batch(img, torch.zeros(1, 1)), train=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("pipeline %.1f ms mem %.1f GB" % ((t1 - t0) * 1e3, torch.cuda.max_memory_allocated() / 1e9), flush=True)
PY
rocprofv3 --kernel-trace -d $OUT/db -o t -- python3 /tmp/inf16.py > $OUT/run.log 2>&1
python3 $R/scratch/r4/kstat.py $OUT/db 30 > $OUT/kstat.txt 2>&1
rm -rf $OUT/db
grep pipeline $OUT/run.log; cat $OUT/kstat.txt
