"""ms per graph-replayed training step: python scratch/r3/step_time.py [batch] [steps] [dtype ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from spr_pick_amd import Denoiser, graph_step, synthetic
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dtypes = sys.argv[3:] or ["f32", "bf16"]
dev = torch.device("cuda:0")
mics = [synthetic.micrograph(i) for i in range(4)]
batches = synthetic.patch_batches(8, batch, mics, seed=100, device=dev)
for dtype in dtypes:
    torch.manual_seed(0)
    den = Denoiser(bench.make_cfg(), device=dev, mode="joint"); den.train()
    if dtype != "f32": den.set_conv_dtype(dtype)
    opt = graph_step.make_adam([p for p in den.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.99))
    st = graph_step.GraphedTrainStep(den, batch, 64, 0.75, 0.01)
    st.prepare(*batches[0])
    np.random.seed(0)
    for rep in range(2):
        for i in range(5): st(*batches[i % 8]); opt.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps): st(*batches[i % 8]); opt.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print("%-5s batch %d  %7.3f ms/step  %7.1f patches/s  kernels/step %s  env SKIP_WT=%s" % (dtype, batch, dt * 1e3, batch / dt, st.kernels_per_step, os.environ.get("SPRK_SKIP_WT", "0")), flush=True)
