#!/usr/bin/env python
"""Headline benchmark: joint denoise+detect TRAINING throughput (patches/s) on synthetic
1024x1024 micrographs, 64x64 patches (BASELINE.json configs[1]: ssdn/gaussian, batch 32 per GPU),
plus the inference legs (filled whole-micrograph forward + NMS, Mpix/s; configs[2] is the 4096^2 one).

  python bench.py --gpus N --steps K --warmup W
  N > 1: either launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (RANK /
  WORLD_SIZE in the environment), or started bare: the parent then — before anything touches the GPU — starts that very
  launcher as a child, relays rank 0's JSON line and exits with the child's code (self_launch).

One step = zero_grad -> Denoiser.run_pipeline(train) (2 JointNetwork passes + sigma net + losses)
-> backward -> [flat gradient all-reduce over RCCL] -> Adam.  Batches are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PATCH_STEP = 67.90e9       # 33.95 GMAC fwd+bwd per patch
FLOP_PER_INFER_PIXEL = 3.4756e6
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
WINOGRAD_GAIN = 2.25                # F(2x2,3x3): 16 multiplies per 2x2 output tile and channel pair instead of 36

# kernel classes of include/sprk.h (sprk_prof_*).  `bound` = the algorithmic-FLOP rate the kernel's OWN algorithm
# allows on the fp32 matrix pipes: a direct convolution issues every algorithmic FLOP as an MFMA FLOP (157.3 TF);
# a Winograd F(2x2,3x3) kernel issues 4/9 of them, so 2.25 x 157.3 TF of algorithmic FLOPs saturate the pipes.
KCLASS = {
    0: ("conv_mfma_kernel<4, 6>", 1.0,
        "direct implicit-GEMM forward / backward-data, all <MT=4, NT=6, row bases, staging> instantiations"),
    1: ("conv_wgrad_mfma_kernel", 1.0, "direct backward-weight, all instantiations"),
    2: ("conv_mfma_kernel<other> + wino_conv_kernel<3>", 1.0,
        "the narrow / small direct instantiations and the 48-channel Winograd kernel (priced as direct)"),
    3: ("wino_conv_kernel<6>", WINOGRAD_GAIN,
        "Winograd F(2x2,3x3) forward / backward-data of the 96-channel 3x3 layers. achieved = algorithmic FLOPs of the "
        "convolution (2*N*H*W*Cout*Cin*9, SURVEY 8d) / launch time; the kernel issues 4/9 of them as MFMA FLOPs, so "
        "peak = 2.25 x 157.3 TF and frac = the busy fraction of the fp32 matrix pipes"),
    4: ("wino_wgrad_kernel", WINOGRAD_GAIN,
        "Winograd F(2x2,3x3) backward-weight of the largest layers (main and tail-channel launches); issues 4/9 of "
        "the algorithmic FLOPs as MFMA FLOPs"),
}


def make_cfg():
    from spr_pick_amd import cfg, params
    c = cfg.base()
    c[params.ConfigValue.ALGORITHM] = params.NoiseAlgorithm.SELFSUPERVISED_DENOISING
    c[params.ConfigValue.NOISE_STYLE] = "gaussian"
    c[params.ConfigValue.NOISE_VALUE] = params.NoiseValue.UNKNOWN_VARIABLE
    c[params.ConfigValue.NMS] = 18
    c[params.ConfigValue.BB] = 24
    return cfg.infer(c, model_only=True)


def kernel_source_hash():
    """Identity of the kernels a committed PMC profile belongs to: SHA-1 over the HIP sources."""
    h = hashlib.sha1()
    for p in sorted(glob.glob(os.path.join(ROOT, "spr_pick_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "spr_pick_amd", "csrc", "*.h"))):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def cpu_share():
    """(threads to use, affinity count, cgroup quota or None).  The GPU box shows 256 logical CPUs in the affinity mask
    but grants one GPU's share of them (16: the pool's rule for worker pools); a torch intra-op pool sized to the mask
    is oversubscribed ~16x and the CPU path becomes orders slower (measured in round 2, and again this round: the
    baseline did not finish in 7 minutes).  So: min(affinity, cgroup cpu.max quota if one is set, 16); SPRK_CPU_THREADS
    overrides.  All three numbers are printed in cpu_baseline."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = max(1, int(round(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    if os.environ.get("SPRK_CPU_THREADS"):
        return int(os.environ["SPRK_CPU_THREADS"]), aff, quota
    return min(aff, quota or aff, 16), aff, quota


def cpu_threads():
    return cpu_share()[0]


def log(msg):
    """Progress to stderr (the JSON line on stdout stays alone): a silent run looks hung to the GPU pool's guard."""
    print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def cpu_baseline(micrographs, seconds):
    """The oracle (CPU restatement of the reference path) timed on the host cores, BASELINE
    configs[0]: batch 4, same synthetic patches; a bounded sample of ~`seconds` of CPU work."""
    from oracle import pipeline, weights
    from spr_pick_amd import synthetic
    threads = cpu_threads()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    sd = weights.make_state(weights.denoiser_shapes(), seed=0)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99))
    batches = synthetic.patch_batches(4, 4, micrographs, seed=1, device="cpu")
    g = torch.Generator().manual_seed(0)

    def step(i):
        inp, tgt = batches[i % len(batches)]
        opt.zero_grad()
        eps = torch.randn(inp.shape, generator=g)
        eps_f = torch.randn(inp.shape, generator=g)
        res = pipeline.joint_pipeline(sd, inp, tgt, 0.75, 0.01, True, eps, eps_f, float(torch.rand(1, generator=g)))
        res["LOSS"].mean().backward()
        opt.step()

    step(0)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        step(n + 1)
        n += 1
        if time.perf_counter() - t0 >= seconds or n >= 64:
            break
    dt = time.perf_counter() - t0
    _, aff, quota = cpu_share()
    return {"value": 4 * n / dt, "unit": "patches/s", "cores": threads, "affinity_cpus": aff, "cgroup_quota_cpus": quota,
            "kind": "port",
            "sample": "%d optimisation steps of batch 4 (64x64 patches, same synthetic micrographs) in %.1f s; "
                      "oracle/ restatement on torch CPU" % (n, dt)}


def cpu_inference_baseline(size):
    """one filled eval forward + NMS of a `size`^2 synthetic micrograph on the host cores
    (oracle/ restatement: torch CPU networks + the C greedy NMS)."""
    from oracle import nms, pipeline, weights
    from spr_pick_amd import synthetic
    threads = cpu_threads()
    torch.set_num_threads(threads)
    sd = weights.make_state(weights.denoiser_shapes(), seed=0)
    img = torch.from_numpy(synthetic.micrograph(7, size=size)[0].astype(np.float32) / 255.0)[None, None]
    eps = torch.randn(img.shape, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        t0 = time.perf_counter()
        res = pipeline.joint_pipeline(sd, img, None, 0, 0, False, eps)
        t1 = time.perf_counter()
        s, _ = nms.nms_c(res["DETECT"][0, 0].numpy(), 18, 0.02)
        t2 = time.perf_counter()
    return {"value": size * size / (t2 - t0) / 1e6, "unit": "Mpix/s", "cores": threads, "kind": "port",
            "sample": "one %dx%d filled eval forward (%.1f s) + C greedy NMS r=18 thr=0.02 (%.3f s, %d picks); oracle/ "
                      "restatement on torch CPU" % (size, size, t1 - t0, t2 - t1, len(s))}


def prof_collect(L, kc):
    n_, ms_, fl_ = ctypes.c_long(), ctypes.c_double(), ctypes.c_double()
    L.sprk_prof_collect(kc, ctypes.byref(n_), ctypes.byref(ms_), ctypes.byref(fl_))
    return n_.value, ms_.value, fl_.value


def pick_agreement(a, b):
    """|A & B| / |A | B| of two coordinate lists [n,2]."""
    sa = set(map(tuple, a.tolist())); sb = set(map(tuple, b.tolist()))
    return len(sa & sb) / max(len(sa | sb), 1)


def inference_leg(den, dev, size, reps, keep_picks=None):
    """Whole-micrograph filled inference as the evaluator runs it (reference train.py:383-415,557-571): host
    uint8 micrograph -> H2D -> /255 -> JointNetwork filled + sigma net + posterior mean + clamped sigmoid -> NMS
    (r=18, thr 0.02) -> picks back on the host.  H2D and the D2H of the picks are inside the timed region."""
    from spr_pick_amd import DetectionDataset, nms_device, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    host_u8 = torch.from_numpy(synthetic.micrograph(7, size=size)[0]).pin_memory()
    zeros = torch.zeros(1, 1)
    gen = torch.Generator(device=dev)
    den.eval(); den.fill()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t_net = t_nms = t_h2d = 0.0
    try:
        with torch.no_grad():
            def once(timed):
                nonlocal t_net, t_nms, t_h2d
                ev[0].record()
                img = host_u8.to(dev, non_blocking=True).to(torch.float32).div_(255.0)[None, None]
                ev[1].record()
                # the reparameterisation noise (joint_network_v2.py:473) is drawn inside the timed region, from a
                # generator re-seeded per micrograph: every leg sees the same noise, so pick lists of
                # different operand precisions can be compared
                gen.manual_seed(7)
                eps = torch.randn((1, 1, size, size), dtype=torch.float32, device=dev, generator=gen)
                oe = den.run_pipeline(DetectionDataset.make_batch(img, zeros), train=False, eps=eps)
                score = oe[P.DETECT][0, 0]
                ev[2].record()
                s, c = nms_device(score, 18, 0.02)
                s, c = s.cpu(), c.cpu()
                ev[3].record()
                torch.cuda.synchronize()
                if timed:
                    t_h2d += ev[0].elapsed_time(ev[1]); t_net += ev[1].elapsed_time(ev[2]); t_nms += ev[2].elapsed_time(ev[3])
                if keep_picks is not None:
                    keep_picks[:] = [c.numpy().copy()]
                return len(s)
            once(False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                n_picks = once(True)
            dt = (time.perf_counter() - t0) / reps
    finally:
        den.unfill(); den.train()
    peak_gb = torch.cuda.max_memory_allocated(dev) / 1e9
    torch.cuda.empty_cache()
    # dominant MFMA kernel of the leg: one more (untimed) repetition with every convolution class bracketed by events
    from spr_pick_amd import _lib
    L = _lib.lib()
    for kc in KCLASS:
        prof_collect(L, kc)
    den.eval(); den.fill()
    try:
        with torch.no_grad():
            L.sprk_prof_enable(31)
            once(False)
            L.sprk_prof_enable(0)
    finally:
        L.sprk_prof_enable(0)
        den.unfill(); den.train()
    cls = {kc: prof_collect(L, kc) for kc in KCLASS}
    dom = max(cls, key=lambda kc: cls[kc][1])
    n_d, ms_d, fl_d = cls[dom]
    conv_ms = sum(v[1] for v in cls.values())
    ach = fl_d / (ms_d * 1e-3) / 1e12 if ms_d > 0 else 0.0
    roof = {"bound": "mfma", "kernel": KCLASS[dom][0], "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS * KCLASS[dom][1],
            "unit": "TFLOP/s", "frac": ach / (PEAK_FP32_MFMA_TFLOPS * KCLASS[dom][1]), "launches": n_d,
            "avg_launch_ms": ms_d / max(n_d, 1), "traffic": None,
            "share_of_network_ms": ms_d / max(t_net / reps, 1e-9),
            "all_conv_classes_ms": {KCLASS[k][0]: round(v[1], 3) for k, v in cls.items()},
            "conv_share_of_network_ms": conv_ms / max(t_net / reps, 1e-9),
            "events_region": "one extra repetition after the timed ones, every MFMA convolution launch bracketed by HIP "
                             "events on its stream (operand precision as in the timed repetitions)"}
    torch.cuda.empty_cache()
    nms_ms = t_nms / reps
    nms_bytes = 4.0 * size * size + 12.0 * n_picks      # read every score once, write the picks
    return {"metric": "inference_mpix_per_sec", "value": size * size / dt / 1e6, "unit": "Mpix/s", "size": [size, size],
            "reps": reps, "ms_per_micrograph": dt * 1e3, "h2d_ms": t_h2d / reps, "network_ms": t_net / reps,
            "includes": "H2D of the uint8 micrograph, network, NMS, D2H of the picks", "picks": int(n_picks),
            "peak_hbm_gb": peak_gb, "roofline": roof,
            "mfma_frac_direct": size * size / dt * FLOP_PER_INFER_PIXEL / (PEAK_FP32_MFMA_TFLOPS * 1e12),
            "nms": {"ms": nms_ms, "radius": 18, "threshold": 0.02, "bytes": nms_bytes,
                    "achieved_gbs": nms_bytes / (nms_ms * 1e-3) / 1e9 if nms_ms > 0 else 0.0,
                    "frac_of_hbm": nms_bytes / (nms_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if nms_ms > 0 else 0.0,
                    "picks_per_s": n_picks / (nms_ms * 1e-3) if nms_ms > 0 else 0.0,
                    "note": "algorithmic bytes 4*H*W + 12*n_picks; the greedy dependency chain, not bandwidth, bounds it "
                            "(includes the D2H of the picks)"}}


def self_launch(gpus):
    """`python bench.py --gpus N` without a launcher around it: start `torch.distributed.run` with N ranks of this script
    as a CHILD process (this process has not touched the GPU and never will: no exec, nothing to tear down), pass its
    stdout (rank 0's JSON line) and stderr through, return its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: %s" % " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, env=env)
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return proc.wait()


def stub_main(args):
    """SPRK_BENCH_STUB=1 (tests/test_bench_launch_cpu.py): the launch / timing / reporting skeleton of main() with a
    stand-in step (a small all-reduce + a sleep) instead of the GPU step — rendezvous, barrier-bracketed timed region,
    max over ranks, one JSON line from rank 0, non-zero exit when a rank fails; runs on CPU under gloo."""
    import torch.distributed as dist
    from spr_pick_amd import distributed
    rank, world, _ = distributed.init_from_env()
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    if os.environ["SPRK_BENCH_STUB"] == "fail%d" % rank:
        raise RuntimeError("stub: rank %d fails on purpose" % rank)
    batch = args.global_batch // world if args.global_batch else args.batch
    buf = torch.ones(1024)

    def step():
        if world > 1:
            dist.all_reduce(buf)
            buf.div_(world)
        time.sleep(0.002 * (1 + rank))            # ranks differ: the report must carry the slowest

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    if rank == 0:
        print(json.dumps({"metric": "train_patches_per_sec", "value": world * batch * args.steps / dt, "unit": "patches/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
                          "dtype": "stub", "data": "none (SPRK_BENCH_STUB)", "config": {"workload": "stub", "per_gpu_batch": batch,
                                                                                      "global_batch": batch * world}}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU per step (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed global batch split over the GPUs (strong scaling; 128 = BASELINE configs[3])")
    ap.add_argument("--graph", choices=("on", "off"), default="on",
                    help="replay the forward+backward from HIP graphs (on) or enqueue every launch from Python (off)")
    ap.add_argument("--event-steps", type=int, default=10,
                    help="eager steps after the timed region whose dominant-kernel launches are bracketed by HIP events")
    ap.add_argument("--dtype", choices=("f32", "bf16", "f16"), default="f32",
                    help="operand precision of the U-Nets' MFMA convolutions in the headline (timed) region")
    ap.add_argument("--also-dtype", choices=("none", "bf16", "f16"), default="bf16",
                    help="second, shorter timed region with this operand precision (reported under train_<dtype>)")
    ap.add_argument("--sustain-seconds", type=float, default=5.0,
                    help="extra back-to-back steps after the timed region until this many seconds (0 = skip)")
    ap.add_argument("--batch16", choices=("on", "off"), default="on",
                    help="also time the step at 16 patches per GPU (BASELINE configs[3]'s per-GPU batch)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-pipeline", choices=("on", "off"), default="on",
                    help="BASELINE configs[4] in small through the `joint` CLI (full_pipeline.py): train on a synthetic set on "
                         "disk, evaluate the final weights, recall / precision of the picks against the planted particles")
    ap.add_argument("--full-pipeline-args", default="--micrographs 16 --iterations 160000 --batch 16 --dtypes f32,mixed16 "
                                                    "--agreement f16 --print-interval 16000")
    ap.add_argument("--infer-size", type=int, default=1024, help="side of the small inference micrograph (0 = skip)")
    ap.add_argument("--infer-large", type=int, default=4096, help="side of the configs[2] micrograph (0 = skip)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if os.environ.get("SPRK_BENCH_STUB"):
        return stub_main(args)

    from spr_pick_amd import Denoiser, _lib, distributed, graph_step, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    import torch.distributed as dist

    rank, world, local = distributed.init_from_env()
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d (a launcher with another --nproc-per-node is around this run)" % (
        world, args.gpus)
    local = local % torch.cuda.device_count()   # (== LOCAL_RANK on a real multi-GPU node)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L = _lib.lib()
    if args.global_batch:
        assert args.global_batch % world == 0, "--global-batch must divide over the GPUs"
        batch, scaling = args.global_batch // world, "strong"
    else:
        batch, scaling = args.batch, "weak"

    mics = [synthetic.micrograph(i) for i in range(4)]
    torch.manual_seed(0)
    den = Denoiser(make_cfg(), device=dev, mode="joint")
    den.train()
    params = [p for p in den.parameters() if p.requires_grad]
    opt = graph_step.make_adam(params, lr=1e-4, betas=(0.9, 0.99))
    use_graph = args.graph != "off"
    if args.dtype != "f32":
        den.set_conv_dtype(args.dtype)
    stepper = graph_step.GraphedTrainStep(den, batch, 64, 0.75, 0.01, world=world, graph=use_graph)
    nb = min(16, args.steps + args.warmup)
    batches = synthetic.patch_batches(nb, batch, mics, seed=100 + rank, device=dev)
    np.random.seed(1000 + rank)
    torch.manual_seed(1000 + rank)

    def step(i, eager=False):
        """forward + backward (HIP-graph replay unless `eager`) -> in-place flat gradient all-reduce -> one-launch Adam"""
        inp, tgt = batches[i % nb]
        o = stepper(inp, tgt, eager=eager)
        stepper.grads.all_reduce(world)
        opt.step()
        return o

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def collect(kc):
        n_, ms_, fl_ = ctypes.c_long(), ctypes.c_double(), ctypes.c_double()
        L.sprk_prof_collect(kc, ctypes.byref(n_), ctypes.byref(ms_), ctypes.byref(fl_))
        return n_.value, ms_.value, fl_.value

    # Kernel timing by HIP events (sprk_prof_*) needs eager launches: events cannot bracket the nodes of a replayed
    # graph.  (1) two eager steps with every MFMA class bracketed pick the class with the most GPU time; (2) the
    # timed region replays the graphs (that is `value`); (3) right after it `--event-steps` eager steps of the SAME
    # kernels on the same batches are bracketed for the dominant class (the roofline leg) and three more for the
    # others.  With --graph off the timed region itself is bracketed, as in round 1.
    step(0, eager=True)   # first touch (code objects, LDS opt-ins, allocator): not part of the class ranking
    fence()
    L.sprk_prof_enable(31)
    for i in range(2):
        step(i, eager=True)
    fence()
    L.sprk_prof_enable(0)
    log("eager warm-up done")
    warm = {kc: collect(kc) for kc in KCLASS}
    DOM = max(KCLASS, key=lambda kc: warm[kc][1])
    stepper.prepare(*batches[0])     # captures both flip-axis graphs (no-op with --graph off)
    for i in range(args.warmup):
        step(i)
    fence()
    if not use_graph:
        L.sprk_prof_enable(1 << DOM)
    launches0 = L.sprk_launch_count()
    t0 = time.perf_counter()
    c0 = time.thread_time()
    for i in range(args.steps):
        o = step(args.warmup + i)
    t_cpu = time.thread_time() - c0           # CPU time of the launching thread (includes spinning on a full queue)
    t_enq = time.perf_counter() - t0          # wall time until the last step was enqueued (the host may run at
                                              # most 8 steps ahead: the pinned upload ring of the labels)
    fence()
    dt = time.perf_counter() - t0
    L.sprk_prof_enable(0)
    log("timed region done: %.1f ms per step" % (dt / args.steps * 1e3))
    launches = L.sprk_launch_count() - launches0
    last_loss = float(torch.mean(o[P.LOSS].detach()))
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # host cost of enqueueing a step, measured on an idle queue (nothing to wait for): 6 steps from a synchronised GPU
    fence()
    te0 = time.perf_counter()
    for i in range(6):
        step(args.warmup + args.steps + i)
    t_host = (time.perf_counter() - te0) / 6
    fence()

    # sustained leg: at least --sustain-seconds of back-to-back steps whatever --steps was (DVFS steady state on record)
    sustained = None
    if args.sustain_seconds > 0:
        n_sus = max(args.steps, int(np.ceil(args.sustain_seconds / (dt / args.steps))))
        fence()
        ts0 = time.perf_counter()
        for i in range(n_sus):
            step(args.warmup + args.steps + i)
        fence()
        dts = time.perf_counter() - ts0
        t = torch.tensor([dts], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dts = float(t.item())
        sustained = {"value": world * batch * n_sus / dts, "unit": "patches/s", "steps": n_sus, "seconds": dts,
                     "ms_per_step": dts / n_sus * 1e3}

    log("sustained leg done")
    # the timed path against the eager path, in this very run: one more batch through the graph replay and through the
    # eager launches of the same stepper — same parameters (no optimiser step in between), same device RNG state, same
    # flip draw — must give the same loss to the last bit (tests/test_gpu_graph_step.py asserts it for every output and
    # the whole flat gradient)
    replay_check = None
    if use_graph:
        inp_c, tgt_c = batches[(args.warmup + args.steps) % nb]
        rng_state = torch.cuda.get_rng_state(dev)
        o_r = stepper(inp_c, tgt_c, flip_p=0.25)
        loss_r = o_r[P.LOSS].detach().clone()
        torch.cuda.set_rng_state(rng_state, dev)
        o_e = stepper(inp_c, tgt_c, flip_p=0.25, eager=True)
        loss_e = o_e[P.LOSS].detach().clone()
        replay_check = {"final_loss_replay": float(loss_r.mean()), "final_loss_eager": float(loss_e.mean()),
                        "bit_identical": bool(torch.equal(loss_r, loss_e))}
        del o_r, o_e

    if use_graph:
        L.sprk_prof_enable(1 << DOM)
        for i in range(args.event_steps):
            step(args.warmup + args.steps + i, eager=True)
        fence()
        L.sprk_prof_enable(0)
    prof = {DOM: collect(DOM)}
    # the other MFMA kernels: three more eager steps, every class bracketed
    L.sprk_prof_enable(31)
    for i in range(3):
        step(args.warmup + args.steps + i, eager=True)
    fence()
    L.sprk_prof_enable(0)
    extra_dom = collect(DOM)
    others = tuple(k for k in KCLASS if k != DOM)
    for kc in others:
        prof[kc] = collect(kc)
    all_ms = extra_dom[1] + sum(prof[k][1] for k in others)
    all_fl = extra_dom[2] + sum(prof[k][2] for k in others)

    log("event legs done")
    value_of = {args.dtype: world * batch * args.steps / dt}
    # second training leg: the same step with 16-bit MFMA operands in the U-Nets (BASELINE configs[4]); new graphs
    second = None
    if args.also_dtype not in ("none", args.dtype):
        n16 = den.set_conv_dtype(args.also_dtype)
        st2 = graph_step.GraphedTrainStep(den, batch, 64, 0.75, 0.01, world=world, graph=use_graph)
        st2.grads = stepper.grads          # same flat gradient buffer (the optimiser's .grad views stay valid)
        st2._compacted = True

        def step2(i):
            inp, tgt = batches[i % nb]
            o2 = st2(inp, tgt)
            st2.grads.all_reduce(world)
            opt.step()
            return o2
        c16_0 = L.sprk_conv16_launch_count()
        st2._eager(0.25)
        c16_per_step = L.sprk_conv16_launch_count() - c16_0
        st2._warm = 0
        st2.prepare(*batches[0])
        for i in range(3):
            step2(i)
        fence()
        k2 = max(20, args.steps // 4)
        t0 = time.perf_counter()
        for i in range(k2):
            o2 = step2(i)
        fence()
        dt2 = time.perf_counter() - t0
        t = torch.tensor([dt2], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt2 = float(t.item())
        second = {"metric": "train_patches_per_sec", "dtype": args.also_dtype, "value": world * batch * k2 / dt2,
                  "unit": "patches/s", "steps": k2, "ms_per_step": dt2 / k2 * 1e3,
                  "final_loss": float(torch.mean(o2[P.LOSS].detach())),
                  "conv_layers_requesting_16bit": n16, "conv_launches_on_16bit_kernels_per_step": c16_per_step,
                  "note": "same step, %s MFMA operands in forward, backward-data and backward-weight of the U-Nets AND %s "
                          "activation / activation-gradient tensors between their layers (fp32 master weights, weight "
                          "gradients, accumulation, optimiser); layers the 16-bit kernels do not cover (planes below 32x32 "
                          "at this batch, the detector, the output convolutions) run on fp32 tensors" % (
                              args.also_dtype, args.also_dtype)}
        # roofline of this leg: its dominant kernels are HBM-bound (fp32 tensors in and out, 16x the matrix rate), so
        # they are priced in algorithmic GB/s: three eager steps, the two 16-bit kernel classes bracketed by events
        for kc in (5, 6):
            prof_collect(L, kc)
        L.sprk_prof_enable((1 << 5) | (1 << 6))
        for i in range(3):
            inp, tgt = batches[i % nb]
            st2(inp, tgt, eager=True)
            st2.grads.all_reduce(world)
            opt.step()
        fence()
        L.sprk_prof_enable(0)
        r16 = {}
        for kc, nm16 in ((5, "conv16_tile_kernel / conv16_mfma_kernel / conv16_head_kernel (forward, backward-data)"),
                         (6, "wgrad16_kernel / wgrad16_1x1_kernel (backward-weight)")):
            n_, ms_, fl_, by_ = (ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double())
            L.sprk_prof_collect_bytes(kc, ctypes.byref(n_), ctypes.byref(ms_), ctypes.byref(fl_), ctypes.byref(by_))
            if ms_.value > 0:
                r16[kc] = {"kernel": nm16, "launches": n_.value, "avg_launch_ms": ms_.value / max(n_.value, 1),
                           "achieved": by_.value / (ms_.value * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": by_.value / (ms_.value * 1e-3) / 1e9 / PEAK_HBM_GBS, "bound": "hbm", "traffic": None,
                           "achieved_tflops": fl_.value / (ms_.value * 1e-3) / 1e12, "ms_per_step": ms_.value / 3}
        if r16:
            # HBM bytes per launch of each class from the committed PMC passes of this command (profiles/collect.sh: separate
            # FETCH_SIZE / WRITE_SIZE runs with --dtype bf16), used only while the kernel sources are the ones profiled
            t16, t16_src = {}, "not measured in this run (PMC counters need rocprofv3: profiles/collect.sh)"
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
                try:
                    with open(path) as f:
                        tj = json.load(f)
                    if tj.get("kernel_source_hash") == kernel_source_hash() and tj.get("kernels_bf16_step"):
                        for kc, pre in ((5, ("conv16_tile_kernel", "conv16_mfma_kernel", "conv16_head_kernel")),
                                        (6, ("wgrad16_kernel", "wgrad16_1x1_kernel"))):
                            fam = [v for k, v in tj["kernels_bf16_step"].items() if k.startswith(pre)]
                            n = sum(v["launches"] for v in fam)
                            if n:
                                t16[kc] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam) / n
                        t16_src = "%s: rocprofv3 PMC passes of bench.py --dtype bf16 on these kernel sources, launch-weighted over the class; not this run" % os.path.relpath(path, ROOT)
                        break
                except (OSError, KeyError, ValueError):
                    pass
            for kc in r16:
                r16[kc]["traffic"] = t16.get(kc)
                r16[kc]["traffic_source"] = t16_src
                r16[kc]["algorithmic_bytes_per_launch"] = r16[kc]["achieved"] * 1e9 * r16[kc]["avg_launch_ms"] * 1e-3
                r16[kc]["frac_of_achievable_hbm"] = r16[kc]["achieved"] / 6300.0
            dom16 = max(r16, key=lambda k: r16[k]["ms_per_step"])
            second["roofline"] = dict(r16[dom16], note="algorithmic bytes = every activation tensor of a launch once AT ITS "
                                      "STORAGE TYPE (16-bit tensors between the U-Nets' layers, fp32 at their borders) / event "
                                      "time of the launches; peak 8 TB/s, 6.3 TB/s is what a copy reaches on this part "
                                      "(frac_of_achievable_hbm); these kernels are bound by LDS operand reads and the "
                                      "un-overlapped phases of a tile, not by HBM (DESIGN 4.4: scratch/r4/c16bench.py)")
            second["other_16bit_kernels"] = [v for k, v in r16.items() if k != dom16]
        value_of[args.also_dtype] = second["value"]
        del o2, st2
        den.set_conv_dtype(args.dtype)

    log("second dtype leg done")
    # batch 16 per GPU (BASELINE configs[3] = 128 patches over 8 GPUs), same step, own graphs
    b16 = None
    if args.batch16 == "on" and batch != 16:
        b16 = {}
        bt16 = synthetic.patch_batches(8, 16, mics, seed=200 + rank, device=dev)
        for dt_name in (args.dtype,) + ((args.also_dtype,) if args.also_dtype not in ("none", args.dtype) else ()):
            den.set_conv_dtype(dt_name)
            st16 = graph_step.GraphedTrainStep(den, 16, 64, 0.75, 0.01, world=world, graph=use_graph)
            st16.grads = stepper.grads
            st16._compacted = True
            st16._warm = 1
            st16.prepare(*bt16[0])

            def step16(i):
                o16 = st16(*bt16[i % 8])
                st16.grads.all_reduce(world)
                opt.step()
                return o16
            for i in range(3):
                step16(i)
            fence()
            k16 = max(40, args.steps // 2)
            t0 = time.perf_counter()
            for i in range(k16):
                step16(i)
            fence()
            d16 = time.perf_counter() - t0
            t = torch.tensor([d16], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d16 = float(t.item())
            ref_rate = value_of[dt_name]
            b16[dt_name] = {"value": world * 16 * k16 / d16, "unit": "patches/s", "steps": k16, "ms_per_step": d16 / k16 * 1e3,
                            "frac_of_batch%d_rate" % batch: world * 16 * k16 / d16 / ref_rate}
            del st16
        den.set_conv_dtype(args.dtype)

    log("batch-16 leg done")
    # the last collective is behind us: every rank leaves the process group NOW, together (a rank that tears RCCL down while
    # rank 0 is still busy with the single-GPU legs would leave rank 0's own teardown waiting for peers that are gone)
    parallelism = ("dp%d (in-place flat fp32 gradient all-reduce over %s, %d floats%s)" % (
                       world, distributed.backend_name(), stepper.grads.live_numel,
                       "; forced 1-rank collective" if world == 1 else "")
                   if stepper.grads.collectives else "single GPU")
    fence()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank != 0:
        return
    infer = infer_large = None
    if rank == 0:
        del o
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        if args.infer_size:
            infer = inference_leg(den, dev, args.infer_size, reps=5)
        log("inference leg done")
        if args.infer_large:
            torch.cuda.reset_peak_memory_stats(dev)
            picks32, picks16 = [], []
            infer_large = inference_leg(den, dev, args.infer_large, reps=3, keep_picks=picks32)
            infer_large["workload"] = "BASELINE configs[2]: one of the 128 synthetic 4096x4096 micrographs per repetition"
            if args.also_dtype != "none":
                # same leg with 16-bit MFMA operands in the U-Nets (fp16 for inference: 8x finer than bf16, range is no
                # issue in the forward pass); picks are then no longer bit-identical to the fp32 path
                den.set_conv_dtype("f16")
                torch.cuda.reset_peak_memory_stats(dev)
                i16 = inference_leg(den, dev, args.infer_large, reps=3, keep_picks=picks16)
                den.set_conv_dtype(args.dtype)
                infer_large["f16_operands"] = {k: i16[k] for k in ("value", "unit", "ms_per_micrograph", "network_ms", "picks",
                                                                    "peak_hbm_gb")}
                infer_large["f16_operands"]["pick_agreement_with_fp32"] = pick_agreement(picks32[0], picks16[0])
                infer_large["f16_operands"]["pick_agreement_note"] = (
                    "|A & B| / |A | B| of the coordinate sets after NMS: same micrograph, same reparameterisation noise, fp16 "
                    "vs fp32 MFMA operands in the U-Nets")

    patches = world * batch * args.steps
    value = patches / dt
    nm, gain, note = KCLASS[DOM]
    n0, ms0, fl0 = prof[DOM]
    achieved = fl0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else 0.0
    peak = PEAK_FP32_MFMA_TFLOPS * gain
    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read in-process.  They are collected with
    # rocprofv3 (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied) on this same command by
    # profiles/collect.sh; the committed summary is used ONLY when it was taken from the kernel sources in this tree.
    traffic, traffic_src = None, "not measured in this run (PMC counters need rocprofv3: profiles/collect.sh)"
    khash = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                tj = json.load(f)
            if tj.get("kernel_source_hash") == khash:
                traffic = tj["kernels"][nm]["hbm_bytes_per_launch"]
                traffic_src = "%s: rocprofv3 PMC passes of this command on these kernel sources (hash %s), not this run" % (
                    os.path.relpath(path, ROOT), khash)
                break
        except (OSError, KeyError, ValueError):
            pass
    out = {
        "metric": "train_patches_per_sec", "value": value, "unit": "patches/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s: ssdn --noise_style gaussian --noise_value var, joint mode, "
                               "64x64 patches from 4 synthetic 1024x1024 micrographs, batch %d per GPU, alpha 0.75, "
                               "tau 0.01, Adam; %s" % (
                                   "BASELINE configs[3] (global batch %d over %d GPUs, data-parallel, RCCL gradient all-reduce)" % (
                                       batch * world, world) if (args.global_batch and world > 1) else
                                   ("BASELINE configs[1] per GPU, weak scaling over %d GPUs" % world if world > 1 else
                                    "BASELINE configs[1]"), batch, "fp32 MFMA convolutions (Winograd F(2x2,3x3) for the wide 3x3 layers)"
                                                    if args.dtype == "f32" else
                                                    "%s-operand MFMA convolutions in the U-Nets where faster, fp32 elsewhere" % args.dtype),
                   "per_gpu_batch": batch, "global_batch": batch * world, "patch": 64,
                   "parallelism": parallelism,
                   "execution": ("forward+backward replayed from 2 HIP graphs (one per flip axis, %d kernels each), eager "
                                 "all-reduce + one-launch Adam (sprk_adam_multi)" % (stepper.kernels_per_step or 0)) if use_graph
                                else "every launch enqueued from Python (--graph off)"},
        "roofline": {"bound": "mfma", "kernel": nm, "kernel_note": note,
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "peak_note": "fp32 MFMA dense peak %.1f TF x %.2f (algorithmic FLOPs per issued MFMA FLOP of this "
                                  "kernel's algorithm)" % (PEAK_FP32_MFMA_TFLOPS, gain),
                     "achieved_vs_direct_conv_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "launches": n0, "avg_launch_ms": ms0 / max(n0, 1),
                     "events_region": ("%d eager steps right after the graph-replayed timed region (same kernels, same "
                                       "batches; HIP events cannot bracket nodes inside a replayed graph)" % args.event_steps)
                                      if use_graph else "the timed region",
                     "gpu_time_share_in_warmup": warm[DOM][1] / max(sum(w[1] for w in warm.values()), 1e-9)},
        "whole_step_mfma_frac": value / world * FLOP_PER_PATCH_STEP / (PEAK_FP32_MFMA_TFLOPS * 1e12),
        "whole_step_note": "all algorithmic FLOPs of a step (67.90 GFLOP/patch) / step time / the fp32 direct-convolution "
                           "MFMA peak",
        "other_mfma_kernels": [{"kernel": KCLASS[k][0], "launches": prof[k][0],
                                "avg_launch_ms": prof[k][1] / max(prof[k][0], 1),
                                "achieved_tflops": prof[k][2] / (prof[k][1] * 1e-3) / 1e12 if prof[k][1] > 0 else 0.0,
                                "frac_of_own_bound": (prof[k][2] / (prof[k][1] * 1e-3) / 1e12 /
                                                      (PEAK_FP32_MFMA_TFLOPS * KCLASS[k][1])) if prof[k][1] > 0 else 0.0}
                               for k in others],
        "other_mfma_note": "three extra steps after the timed region with every MFMA launch bracketed by events",
        "all_conv_mfma_tflops": all_fl / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0,
        "kernel_launches_per_step": (stepper.kernels_per_step if use_graph else launches / args.steps),
        "host_launch_calls_per_step": launches / args.steps,
        "host_enqueue_ms_per_step": t_host * 1e3,
        "host_enqueue_note": "wall time per step until the launching thread has enqueued it, from an idle queue (6 steps)",
        "host_cpu_ms_per_step": t_cpu / args.steps * 1e3,
        "host_enqueue_wall_ms_per_step": t_enq / args.steps * 1e3,
        "final_loss": last_loss, "kernel_source_hash": khash,
        "graph_fallback": stepper.fallback_reason,
    }
    if sustained:
        out["sustained_patches_per_sec"] = sustained["value"]
        out["sustained"] = sustained
    if replay_check:
        out.update(replay_check)
    if b16:
        out["batch16"] = b16
    if second:
        out["train_" + second["dtype"]] = second
    if infer:
        out["inference"] = infer
    if infer_large:
        out["inference_large"] = infer_large
    if world == 1 and args.full_pipeline == "on":
        # the whole workflow through the CLI, with the trainer's own loop (device patch feed, sampler, logging, checkpoints)
        # and the evaluator's (file reading, PNG / score writers): what a user of `joint train start` / `joint eval` gets
        import full_pipeline
        del stepper, opt, den
        torch.cuda.empty_cache()
        fp = full_pipeline.main(args.full_pipeline_args.split(), quiet=True)
        for r in fp["runs"].values():
            b = r["train"]["batch"]
            resident = {32: value_of, 16: {k: v["value"] for k, v in (b16 or {}).items()}}.get(b, {})
            key = {"mixed16": "bf16"}.get(r["dtype"], r["dtype"])
            if resident.get(key):
                r["train"]["resident_batch_patches_per_s"] = resident[key]
                r["train"]["trainer_loop_vs_resident_batches"] = r["train"]["trainer_loop_patches_per_s"] / resident[key]
        out["full_pipeline"] = fp
        log("full pipeline leg done")
    log("GPU legs done")
    if not args.no_cpu_baseline:       # rank 0, after the last barrier; at N > 1 too (the other ranks have left)
        out["cpu_baseline"] = cpu_baseline(mics, args.cpu_seconds)
        log("CPU training baseline done")
        out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if args.infer_size:
            out["cpu_baseline"]["inference"] = cpu_inference_baseline(args.infer_size)
            out["inference"]["vs_cpu_baseline"] = infer["value"] / out["cpu_baseline"]["inference"]["value"]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
