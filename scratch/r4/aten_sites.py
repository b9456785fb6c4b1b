"""Which lines of the package launch the small ATen kernels of a training step?  One eager step under torch.profiler
with Python stacks; every ATen operator that launched a kernel is attributed to its innermost frame inside spr_pick_amd
(forward) or to the autograd node that ran it (backward).   python scratch/r4/aten_sites.py [dtype] [batch]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
import bench
from spr_pick_amd import Denoiser, graph_step, synthetic
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
mics = [synthetic.micrograph(i) for i in range(2)]
torch.manual_seed(0)
den = Denoiser(bench.make_cfg(), device=dev, mode="joint"); den.train()
if dt != "f32": den.set_conv_dtype(dt)
opt = graph_step.make_adam([p for p in den.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.99))
st = graph_step.GraphedTrainStep(den, batch, 64, 0.75, 0.01, world=1, graph=False)
bs = synthetic.patch_batches(4, batch, mics, seed=100, device=dev)
for i in range(3):
    st(*bs[i], eager=True); opt.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    st(*bs[3], eager=True, flip_p=0.25); opt.step()
    torch.cuda.synchronize()
ev = prof.events()
sites = collections.Counter(); total = 0
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels: continue
    if not e.name.startswith("aten::") and "Memcpy" not in e.name and "Memset" not in e.name: continue
    if any(c.kernels for c in e.cpu_children): continue          # attribute to the innermost operator
    frame = next((f for f in (e.stack or []) if "spr_pick_amd" in f), None)
    if frame is None:
        p = e.cpu_parent
        while p is not None and not ("Backward" in p.name or "autograd::engine" in p.name): p = p.cpu_parent
        frame = "backward of " + (p.name if p is not None else "?")
        q = e.cpu_parent
        if q is not None and q.name.startswith("aten::"): frame += " / " + q.name
    sites[(e.name, frame.replace(ROOT + "/", ""))] += len(e.kernels); total += len(e.kernels)
print("ATen-launched kernels in one eager step (%s, batch %d): %d" % (dt, batch, total))
for (name, frame), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%3d  %-28s %s" % (n, name, frame[:150]))
