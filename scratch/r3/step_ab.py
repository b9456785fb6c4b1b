"""Same-box A/B of the graph-replayed training step: python scratch/r3/step_ab.py [batch] [steps] -- prints ms/step for
f32 and bf16 with networks.FUSE_ACT_BWD on/off."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from spr_pick_amd import Denoiser, graph_step, networks, synthetic

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
mics = [synthetic.micrograph(i) for i in range(4)]
batches = synthetic.patch_batches(8, batch, mics, seed=100, device=dev)


def run(dtype, fuse):
    networks.FUSE_ACT_BWD = fuse
    torch.manual_seed(0)
    den = Denoiser(bench.make_cfg(), device=dev, mode="joint")
    den.train()
    if dtype != "f32":
        den.set_conv_dtype(dtype)
    opt = graph_step.make_adam([p for p in den.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.99))
    st = graph_step.GraphedTrainStep(den, batch, 64, 0.75, 0.01)
    st.prepare(*batches[0])
    np.random.seed(0)
    for i in range(5):
        st(*batches[i % 8]); opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        o = st(*batches[i % 8]); opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%-5s fuse=%d  %7.3f ms/step  %7.1f patches/s  kernels/step %s" % (dtype, fuse, dt * 1e3, batch / dt, st.kernels_per_step), flush=True)


for dtype in ("f32", "bf16"):
    for fuse in (False, True, False, True):
        run(dtype, fuse)
