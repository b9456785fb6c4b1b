#!/usr/bin/env python
"""Headline benchmark: joint denoise+detect TRAINING throughput (patches/s) on synthetic
1024x1024 micrographs, 64x64 patches (BASELINE.json configs[1]: ssdn/gaussian, batch 32 per GPU).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = zero_grad -> Denoiser.run_pipeline(train) (2 JointNetwork passes + sigma net + losses)
-> backward -> [flat gradient all-reduce over RCCL] -> Adam.  Batches are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
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


def make_cfg():
    from spr_pick_amd import cfg, params
    c = cfg.base()
    c[params.ConfigValue.ALGORITHM] = params.NoiseAlgorithm.SELFSUPERVISED_DENOISING
    c[params.ConfigValue.NOISE_STYLE] = "gaussian"
    c[params.ConfigValue.NOISE_VALUE] = params.NoiseValue.UNKNOWN_VARIABLE
    c[params.ConfigValue.NMS] = 18
    c[params.ConfigValue.BB] = 24
    return cfg.infer(c, model_only=True)


def cpu_baseline(micrographs, seconds):
    """The oracle (CPU restatement of the reference path) timed on the host cores, BASELINE
    configs[0]: batch 4, same synthetic patches; a bounded sample of ~`seconds` of CPU work."""
    from oracle import pipeline, weights
    from spr_pick_amd import synthetic
    # host cores this process may actually use (the GPU box exposes 256 logical CPUs but grants a
    # share of them; oversubscribing torch's intra-op pool makes the CPU path orders slower)
    threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("SPRK_CPU_THREADS", "16")))
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
    return {"value": 4 * n / dt, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": "%d optimisation steps of batch 4 (64x64 patches, same synthetic micrographs) in %.1f s; "
                      "oracle/ restatement on torch CPU" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--infer-size", type=int, default=1024, help="side of the inference micrograph (0 = skip)")
    args = ap.parse_args()

    from spr_pick_amd import Denoiser, DetectionDataset, _lib, distributed, nms_device, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    import torch.distributed as dist

    rank, world, local = distributed.init_from_env()
    assert world == args.gpus, "launch with --nproc-per-node == --gpus (WORLD_SIZE=%d, --gpus %d)" % (world, args.gpus)
    local = local % torch.cuda.device_count()   # (== LOCAL_RANK on a real multi-GPU node)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L = _lib.lib()

    mics = [synthetic.micrograph(i) for i in range(4)]
    torch.manual_seed(0)
    den = Denoiser(make_cfg(), device=dev, mode="joint")
    den.train()
    params = [p for p in den.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99))
    sync = distributed.FlatGradAllReduce(params, world)
    nb = min(16, args.steps + args.warmup)
    batches = synthetic.patch_batches(nb, args.batch, mics, seed=100 + rank, device=dev)
    np.random.seed(1000 + rank)
    torch.manual_seed(1000 + rank)

    def step(i):
        inp, tgt = batches[i % nb]
        opt.zero_grad(set_to_none=True)
        o = den.run_pipeline(DetectionDataset.make_batch(inp, tgt), 0.75, 0.01, train=True)
        torch.mean(o[P.LOSS]).backward()
        sync()
        opt.step()
        return o

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import ctypes

    def collect(kc):
        n_, ms_, fl_ = ctypes.c_long(), ctypes.c_double(), ctypes.c_double()
        L.sprk_prof_collect(kc, ctypes.byref(n_), ctypes.byref(ms_), ctypes.byref(fl_))
        return n_.value, ms_.value, fl_.value

    # kernel classes of include/sprk.h; the roofline leg goes to the class that takes the most GPU time in the
    # warm-up steps (all classes bracketed there; an event pair is not free, so the timed region brackets one)
    names = {0: "conv_mfma_kernel<4, 6>", 1: "conv_wgrad_mfma_kernel", 2: "conv_mfma_kernel<other> + wino_conv_kernel<3>",
             3: "wino_conv_kernel<6>", 4: "wino_wgrad_kernel"}
    notes = {
        0: "direct implicit-GEMM forward / backward-data, all <MT=4, NT=6, row bases, staging> instantiations",
        1: "direct backward-weight, all instantiations",
        2: "the narrow / small direct instantiations and the 48-channel Winograd kernel",
        3: "Winograd F(2x2,3x3) forward / backward-data of the 96-channel 3x3 layers. achieved = algorithmic FLOPs of "
           "the convolution (2*N*H*W*Cout*Cin*9, SURVEY 8d) / launch time; the kernel issues 4/9 of them as MFMA "
           "FLOPs (mfma_pipe_frac = achieved * 4/9 / peak)",
        4: "Winograd F(2x2,3x3) backward-weight of the largest layers (main and tail-channel launches); issues 4/9 "
           "of the algorithmic FLOPs as MFMA FLOPs",
    }
    L.sprk_prof_enable(31)
    for i in range(args.warmup):
        step(i)
    fence()
    L.sprk_prof_enable(0)
    warm = {kc: collect(kc) for kc in names}
    DOM = max(names, key=lambda kc: warm[kc][1]) if args.warmup > 0 else 3
    L.sprk_prof_enable(1 << DOM)   # events around the dominant kernel class only
    launches0 = L.sprk_launch_count()
    t0 = time.perf_counter()
    for i in range(args.steps):
        o = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    L.sprk_prof_enable(0)
    launches = L.sprk_launch_count() - launches0
    last_loss = float(torch.mean(o[P.LOSS].detach()))
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    prof = {DOM: (names[DOM],) + collect(DOM)}
    # the other MFMA kernels: three more steps after the timed region, every class bracketed
    L.sprk_prof_enable(31)
    for i in range(3):
        step(args.warmup + args.steps + i)
    fence()
    L.sprk_prof_enable(0)
    extra_dom = collect(DOM)
    others = tuple(k for k in names if k != DOM)
    for kc in others:
        prof[kc] = (names[kc],) + collect(kc)
    all_ms = extra_dom[1] + sum(prof[k][2] for k in others)
    all_fl = extra_dom[2] + sum(prof[k][3] for k in others)

    infer = None
    if args.infer_size and rank == 0:
        S = args.infer_size
        img = torch.from_numpy(synthetic.micrograph(7, size=S)[0].astype(np.float32) / 255.0).to(dev)[None, None]
        den.eval(); den.fill()
        with torch.no_grad():
            def infer_once():
                oe = den.run_pipeline(DetectionDataset.make_batch(img, torch.zeros(1, 1)), train=False)
                return nms_device(oe[P.DETECT][0, 0], 18, 0.02)
            infer_once()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                s_, c_ = infer_once()
            torch.cuda.synchronize()
            ti = (time.perf_counter() - t1) / reps
        den.unfill(); den.train()
        infer = {"metric": "inference_mpix_per_sec", "value": S * S / ti / 1e6, "unit": "Mpix/s", "size": [S, S],
                 "ms_per_micrograph": ti * 1e3, "picks": int(len(s_)), "nms_radius": 18,
                 "mfma_frac": S * S / ti * FLOP_PER_INFER_PIXEL / (PEAK_FP32_MFMA_TFLOPS * 1e12)}

    if rank != 0:
        return
    patches = world * args.batch * args.steps
    value = patches / dt
    nm, n0, ms0, fl0 = prof[DOM]
    achieved = fl0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else 0.0
    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read in-process; they are
    # collected with rocprofv3 (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied) on
    # this same command and committed under profiles/ 
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            traffic = json.load(f)["kernels"][nm]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": "train_patches_per_sec", "value": value, "unit": "patches/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: ssdn --noise_style gaussian --noise_value var, joint mode, "
                               "64x64 patches from 4 synthetic 1024x1024 micrographs, batch %d per GPU, alpha 0.75, "
                               "tau 0.01, Adam; fp32 MFMA convolutions (Winograd F(2x2,3x3) for the wide 3x3 layers)" % args.batch,
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "patch": 64,
                   "parallelism": "dp%d (flat fp32 grad all-reduce, %d floats)" % (world, sync.numel()) if world > 1 else "single GPU"},
        "roofline": dict({"bound": "mfma", "kernel": nm, "kernel_note": notes[DOM],
                          "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                          "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                          "launches": n0, "avg_launch_ms": ms0 / max(n0, 1),
                          "gpu_time_share_in_warmup": warm[DOM][1] / max(sum(w[1] for w in warm.values()), 1e-9)},
                         **({"mfma_pipe_frac": achieved * 4.0 / 9.0 / PEAK_FP32_MFMA_TFLOPS} if DOM in (3, 4) else {})),
        "whole_step_mfma_frac": value / world * FLOP_PER_PATCH_STEP / (PEAK_FP32_MFMA_TFLOPS * 1e12),
        "other_mfma_kernels": [{"kernel": prof[k][0], "launches": prof[k][1], "avg_launch_ms": prof[k][2] / max(prof[k][1], 1),
                                "achieved_tflops": prof[k][3] / (prof[k][2] * 1e-3) / 1e12 if prof[k][2] > 0 else 0.0}
                               for k in others],
        "other_mfma_note": "three extra steps after the timed region with every MFMA launch bracketed by events",
        "all_conv_mfma_tflops": all_fl / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0,
        "kernel_launches_per_step": launches / args.steps, "final_loss": last_loss,
    }
    if infer:
        out["inference"] = infer
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(mics, args.cpu_seconds)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
