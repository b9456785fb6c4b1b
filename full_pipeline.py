#!/usr/bin/env python
"""BASELINE configs[4] on the GPU(s) at hand: the reference's whole workflow (README.md:79-92) through THIS
package's `joint` CLI — `train start` for --iterations images, then `eval` of the final weights over the
micrograph set — on a synthetic EMPIAR-style set written to disk (spr_pick_amd.synthetic.write_dataset), and
what the reference validates by eye (README.md:88-92) as numbers: recall / precision / average precision of the
written picks against the PLANTED particle centres (a pick within --bb/2 px of a still-unmatched centre).

  python full_pipeline.py --micrographs 512 --iterations 80000 --batch 32 --dtypes f32,f16 --out result.json

Per operand type: one training run, one evaluation of its own checkpoint; with two types also the fp32-trained
checkpoint evaluated with the 16-bit kernels (pick agreement of a TRAINED detector).  Prints one JSON object.
Nothing here is timed with the inputs resident: wall clocks include file reading, the patch feed, logging,
checkpoints and the PNG / score files, as a user of the CLI sees them; the network/NMS share is reported beside them."""
import argparse
import glob
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def read_truth(path):
    out = {}
    with open(path) as f:
        f.readline()
        for line in f:
            n, x, y = line.rstrip("\n").split("\t")
            out.setdefault(n, []).append((int(x), int(y)))
    return {k: np.asarray(v, dtype=np.int64) for k, v in out.items()}


def score_dir_metrics(score_dir, truth, radius, thresholds=(0.13, 0.5)):
    from spr_pick_amd import picks
    per_image, found = [], {}
    for path in sorted(glob.glob(os.path.join(score_dir, "*_scores.txt"))):
        name = os.path.basename(path)[:-len("_scores.txt")]
        _, xy, s = picks.read_scores(path)
        found[name] = (xy, s)
        per_image.append((xy, s, truth.get(name, np.zeros((0, 2)))))
    m = picks.detection_metrics(per_image, radius, thresholds)
    m["micrographs"] = len(per_image)
    return m, found


def agreement(a, b):
    """|A & B| / |A | B| over (micrograph, x, y) of two pick sets, per score threshold."""
    out = {}
    for thr in (0.02, 0.13, 0.5):
        sa = {(n, int(x), int(y)) for n, (xy, s) in a.items() for (x, y), v in zip(xy, s) if v > thr}
        sb = {(n, int(x), int(y)) for n, (xy, s) in b.items() for (x, y), v in zip(xy, s) if v > thr}
        out[str(thr)] = {"jaccard": len(sa & sb) / max(len(sa | sb), 1), "a": len(sa), "b": len(sb)}
    return out


def loss_curve(run_dir):
    rows = {}
    path = os.path.join(run_dir, "metrics.tsv")
    if os.path.exists(path):
        for line in open(path):
            tag, it, v = line.rstrip("\n").split("\t")
            rows.setdefault(tag, []).append((int(it), float(v)))
    return rows


def patch_level_detection(wt, ds, dtype, n_per_class=256, seed=0):
    """Does the DETECTOR itself separate particles from background?  Unfilled eval-mode forward (running BatchNorm
    statistics, the training geometry: 64x64 patches) on patches centred on planted particles and on background positions
    of the evaluation set: AUC of the scores + recall / false-positive rate at the exporter's threshold 0.13.  Independent
    of how the blind-spot U-Net's output level carries over from 64x64 training patches to whole micrographs, which the
    pick-level figures depend on as well (full pipeline)."""
    import torch
    from spr_pick_amd import DetectionDataset, checkpoint, micrograph_io
    from spr_pick_amd.denoiser import Denoiser
    from spr_pick_amd.params import PipelineOutput as P
    truth = read_truth(ds["truth"])
    rows = micrograph_io.read_image_table(ds["images"])[:8]
    den = Denoiser.from_state_dict(checkpoint.load(wt), mode="joint", device="cuda:0")
    if dtype != "f32":
        den.set_conv_dtype(dtype)
    den.eval(); den.unfill()
    rng = np.random.default_rng(seed)
    pos, neg = [], []
    for _, name, path in rows:
        img = micrograph_io.to_unit_float(micrograph_io.load_image(path)).T      # tensors enter transposed: row = x
        cen = truth[name]
        for x, y in cen[rng.permutation(len(cen))[:n_per_class // len(rows)]]:
            pos.append(img[x - 31:x + 33, y - 31:y + 33])      # the 63-wide detector window of a 64 patch is centred on pixel 31
        k = 0
        while k < n_per_class // len(rows):
            x, y = rng.integers(80, img.shape[0] - 80, size=2)
            if ((cen - (x, y)) ** 2).sum(axis=1).min() > 24 ** 2:
                neg.append(img[x - 31:x + 33, y - 31:y + 33]); k += 1
    scores = []
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    with torch.no_grad():
        allp = torch.from_numpy(np.stack(pos + neg)[:, None].astype(np.float32)).cuda()
        for i in range(0, len(allp), 64):
            xb = allp[i:i + 64].contiguous()
            eps = torch.randn(xb.shape, device="cuda:0", generator=gen)
            o = den.run_pipeline(DetectionDataset.make_batch(xb, torch.zeros(len(xb), 1)), train=False, eps=eps)
            scores.append(o[P.DETECT].reshape(-1).float().cpu().numpy())
    sc = np.concatenate(scores)
    sp, sn = sc[:len(pos)], sc[len(pos):]
    auc = float((sp[:, None] > sn[None, :]).mean() + 0.5 * (sp[:, None] == sn[None, :]).mean())
    del den
    torch.cuda.empty_cache()
    return {"patches": [len(pos), len(neg)], "auc": auc, "recall_at_0.13": float((sp > 0.13).mean()),
            "false_positive_rate_at_0.13": float((sn > 0.13).mean()), "median_score_particle": float(np.median(sp)),
            "median_score_background": float(np.median(sn))}


def train_and_eval(ds, work, dtype, args, eval_ds=None):
    """-> (result dict, path of the final weights, picks of the evaluation)"""
    import torch
    from spr_pick_amd import cli
    from spr_pick_amd.params import ConfigValue
    runs = os.path.join(work, "runs_" + dtype)
    os.environ["SPRK_CONV_DTYPE"] = dtype
    argv = ("train start -a ssdn -n gaussian --noise_value var -t %s -l %s -ap %s -tau %s -iter %d --train_batch_size %d "
            "--nms 18 --bb 24 --runs_dir %s --print_interval %d --checkpoint_interval %d --eval_interval %d" % (
                ds["images"], ds["labels"], args.alpha, args.tau, args.iterations, args.batch, runs,
                args.print_interval, args.iterations, args.iterations)).split()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer = cli.start(argv)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    run = trainer.run_dir_path
    timing = dict(getattr(trainer, "timing", {}))
    wt = os.path.join(run, "final-%s.wt" % trainer.denoiser.config_name())
    curve = loss_curve(run)
    res = {"dtype": dtype, "train": {
        "wall_s": t_train, "iterations": args.iterations, "batch": args.batch, "optimiser_steps": args.iterations // args.batch,
        "loop_s": timing.get("loop_s"), "setup_s": timing.get("setup_s"),
        "trainer_loop_patches_per_s": (args.iterations / timing["loop_s"]) if timing.get("loop_s") else None,
        "whole_command_patches_per_s": args.iterations / t_train,
        "step_execution": timing.get("execution"),
        "loss_first": {k: v[1][1] if len(v) > 1 else None for k, v in curve.items() if k.startswith("train/") and "rate" not in k},
        "loss_last": {k: v[-1][1] for k, v in curve.items() if k.startswith("train/") and "rate" not in k},
        "loss_curve": [(it, round(v, 5)) for it, v in curve.get("train/loss", [])][::max(1, len(curve.get("train/loss", [])) // 24)],
        "detect_loss_curve": [(it, round(v, 5)) for it, v in curve.get("train/detect_loss", [])][::max(1, len(curve.get("train/detect_loss", [])) // 24)],
    }}
    del trainer
    torch.cuda.empty_cache()
    ev, found = evaluate(wt, eval_ds or ds, work, dtype, args, tag="own")
    res["eval"] = ev
    res["patch_level_detection"] = patch_level_detection(wt, eval_ds or ds, dtype)
    return res, wt, found


def evaluate(wt, ds, work, dtype, args, tag):
    import torch
    from spr_pick_amd import cli
    os.environ["SPRK_CONV_DTYPE"] = dtype
    runs = os.path.join(work, "eval_%s_%s" % (dtype, tag))
    truth = read_truth(ds["truth"])
    argv = ["eval", "--model", wt, "--dataset", ds["images"], "--runs_dir", runs, "--num", str(ds["n"]), "--nms", "18"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = cli.start(argv)
    torch.cuda.synchronize()
    t_eval = time.perf_counter() - t0
    timing = dict(getattr(ev, "timing", {}))
    m, found = score_dir_metrics(os.path.join(ev.run_dir_path, "eval_imgs"), truth, args.match_radius)
    pix = ds["n"] * ds["size"] * ds["size"]
    out = {"dtype": dtype, "wall_s": t_eval, "micrographs": ds["n"], "mpix_per_s_whole_command": pix / t_eval / 1e6,
           "loop_s": timing.get("eval_loop_s"), "load_s": timing.get("eval_load_s"),
           "mpix_per_s_eval_loop": pix / timing["eval_loop_s"] / 1e6 if timing.get("eval_loop_s") else None,
           "device_s": timing.get("eval_device_s"),
           "mpix_per_s_network_nms": pix / timing["eval_device_s"] / 1e6 if timing.get("eval_device_s") else None,
           "picks_vs_planted_centres": m}
    del ev
    torch.cuda.empty_cache()
    return out, found


def main(argv=None, quiet=False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--micrographs", type=int, default=512)
    ap.add_argument("--train-micrographs", type=int, default=0, help="train on the first K micrographs (0 = all)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--format", default="mrc", choices=("mrc", "png"),
                    help="micrograph files: float32 MRC (loaded min-max-scaled to [0, 1]) or 8-bit PNG of the standardised "
                         "image (loaded as [-3, 3], background 0)")
    ap.add_argument("--iterations", type=int, default=80000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--alpha", type=float, default=0.75)
    ap.add_argument("--tau", type=float, default=0.01)
    ap.add_argument("--dtypes", default="f32,mixed16",
                    help="f32 | bf16 | f16 | mixed16 (bf16 operands in training, fp16 in inference: networks.set_conv_dtype)")
    ap.add_argument("--agreement", default="f16",
                    help="evaluate the fp32-trained checkpoint once more with these MFMA operands and report the agreement of "
                         "the two pick sets (none = skip)")
    ap.add_argument("--print-interval", type=int, default=3200)
    ap.add_argument("--match-radius", type=float, default=12.0, help="--bb / 2")
    ap.add_argument("--work", default=None)
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    work = args.work or tempfile.mkdtemp(prefix="sprk_full_")
    from spr_pick_amd import synthetic
    t0 = time.perf_counter()
    ds = synthetic.write_dataset(os.path.join(work, "set"), args.micrographs, size=args.size, fmt=args.format)
    train_ds = ds
    if args.train_micrographs and args.train_micrographs < args.micrographs:
        train_ds = synthetic.write_dataset(os.path.join(work, "set_train"), args.train_micrographs, size=args.size, fmt=args.format)
    out = {"workload": "BASELINE configs[4] on %s GPU(s): joint train start (%d iterations = images, batch %d, alpha %s, tau %s, "
                       "nms 18, bb 24) on %d synthetic %dx%d micrographs (%d labelled of %d planted particles), then joint "
                       "eval of the final weights on %d micrographs" % (
                           os.environ.get("WORLD_SIZE", "1"), args.iterations, args.batch, args.alpha, args.tau, train_ds["n"],
                           args.size, args.size, train_ds["labelled"], train_ds["planted"], ds["n"]),
           "micrograph_files": ("float32 MRC, loaded min-max-scaled to [0, 1] (utils/loader.py:49-59)" if args.format == "mrc" else
                                "8-bit PNG of the standardised image, loaded as [-3, 3] (utils/loader.py:72-82)"),
           "dataset_write_s": time.perf_counter() - t0, "runs": {}}
    picks_by, wts = {}, {}
    for dtype in args.dtypes.split(","):
        res, wt, found = train_and_eval(train_ds, work, dtype, args, eval_ds=ds)
        out["runs"][dtype] = res
        picks_by[dtype], wts[dtype] = found, wt
        print("[full_pipeline] %s done: train %.1f s, eval %.1f s, AP %.3f" % (
            dtype, res["train"]["wall_s"], res["eval"]["wall_s"], res["eval"]["picks_vs_planted_centres"]["average_precision"]),
            file=sys.stderr, flush=True)
    if "f32" in wts and args.agreement != "none":
        other = args.agreement
        ev, found = evaluate(wts["f32"], ds, work, other, args, tag="f32ckpt")
        out["fp32_checkpoint_evaluated_with_%s_operands" % other] = ev
        out["pick_agreement_with_fp32"] = dict(agreement(picks_by["f32"], found),
                                               note="same fp32-TRAINED checkpoint, same micrographs and noise streams; fp32 vs %s "
                                                    "MFMA operands in the U-Nets; |A & B| / |A | B| of (micrograph, x, y) over the "
                                                    "picks above each score threshold" % other)
    os.environ.pop("SPRK_CONV_DTYPE", None)
    s = json.dumps(out)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s + "\n")
    if not quiet:
        print(s)
    return out


if __name__ == "__main__":
    main()
