"""GPU tests of the rows next to the hot path: the device patch feed against
PIL's crop semantics, and the `joint` CLI counterpart end to end (start -> resume -> eval)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pil_patch(img, x, y, P, flip):
    """What MicrographDataset.__getitem__ produces for one item (datasets/micrograph.py:87-120):
    PIL crop (zero fill), optional FLIP_LEFT_RIGHT, to_tensor, CWH->CHW permute."""
    from PIL import Image
    pil = Image.fromarray(img)
    c = pil.crop((x - P // 2, y - P // 2, x - P // 2 + P, y - P // 2 + P))
    if flip:
        c = c.transpose(Image.FLIP_LEFT_RIGHT)
    a = np.array(c)
    t = a.astype(np.float32) / np.float32(255) if a.dtype == np.uint8 else a.astype(np.float32)
    return t.T


@pytest.mark.parametrize("kind", ["u8", "f32"])
def test_gather_patches_matches_pil_crop(kind):
    from spr_pick_amd import feed
    rng = np.random.RandomState(1)
    shapes = [(300, 340), (256, 256), (330, 290)]
    if kind == "u8":
        imgs = [rng.randint(0, 256, size=s).astype(np.uint8) for s in shapes]
    else:
        imgs = [rng.randn(*s).astype(np.float32) for s in shapes]
    groups = [[(im, np.ones(im.shape, np.uint8), np.zeros(im.shape, np.float32)) for im in imgs]]
    names = [["a", "b", "c"]]
    f = feed.PatchFeed(groups, names, batch=8, patch=64, device="cuda", seed=0)
    assert f.dtype == (0 if kind == "u8" else 1)
    items = []
    for k, (r, c) in enumerate(shapes):
        # interior, every border and corner (crop boxes reaching outside the image are zero filled)
        for x, y in ((c // 2, r // 2), (0, 0), (c - 1, r - 1), (5, r - 3), (c - 2, 7), (31, 32), (c - 32, r - 33)):
            for flip in (0, 1):
                items.append((k, x, y, flip))
    items = np.array(items, dtype=np.int32)
    out = f.gather(items).cpu().numpy()
    assert out.shape == (len(items), 1, 64, 64)
    for b, (k, x, y, flip) in enumerate(items):
        np.testing.assert_array_equal(out[b, 0], _pil_patch(imgs[k], int(x), int(y), 64, int(flip)), err_msg=str(items[b]))
    bad = np.array([[7, 10, 10, 0]], dtype=np.int32)             # unknown micrograph index -> zero patch, no fault
    assert float(f.gather(bad).abs().max()) == 0.0
    f128 = feed.PatchFeed(groups, names, batch=2, patch=128, device="cuda", seed=0)
    o = f128.gather(items[:4]).cpu().numpy()
    for b, (k, x, y, flip) in enumerate(items[:4]):
        np.testing.assert_array_equal(o[b, 0], _pil_patch(imgs[k], int(x), int(y), 128, int(flip)))


def test_patch_feed_batch_layout():
    from spr_pick_amd import feed
    from spr_pick_amd.datasets import DetectionDataset
    rng = np.random.RandomState(2)
    im = rng.randint(0, 256, size=(320, 320)).astype(np.uint8)
    from spr_pick_amd import coordinates
    xs, ys = np.array([100, 150]), np.array([90, 160])
    mask = coordinates.as_mask(im.shape, xs, ys, [3, 3])
    hm = coordinates.as_gaussian(im.shape, xs, ys, bb=24)
    f = feed.PatchFeed([[(im, mask, hm)]], [["m"]], batch=32, patch=64, device="cuda", seed=5)
    data = f.next_batch()
    inp, target = data[DetectionDataset.INPUT], data[DetectionDataset.TARGET]
    assert inp.shape == (32, 1, 64, 64) and inp.is_cuda and target.shape == (32, 1) and not target.is_cuda
    md = data[DetectionDataset.METADATA]
    assert md[DetectionDataset.Metadata.NAME] == ["m"] * 32
    idx = md[DetectionDataset.Metadata.INDEXES]
    for b in range(32):
        coord = int(idx[b]) & 0xFFFFFFFF
        assert float(target[b, 0]) == float(hm.ravel()[coord])
        # centre pixel of the (possibly flipped) transposed patch is the sampled pixel or its mirror neighbour
        y, x = coord // 320, coord % 320
        centre = float(inp[b, 0, 32, 32])
        assert centre in (im[y, x] / np.float32(255), im[y, x - 1] / np.float32(255))


def _write_set(root, n=2, size=320, seed=0):
    from spr_pick_amd import micrograph_io, synthetic
    lines, labels = ["image_name\tpath"], ["image_name\tx_coord\ty_coord"]
    for k in range(n):
        q, centres, _ = synthetic.micrograph(k, size=size, blobs=14, seed=seed)
        path = os.path.join(root, "mic%d.mrc" % k)
        with open(path, "wb") as f:
            micrograph_io.write_mrc(f, q.astype(np.float32))
        lines.append("mic%d\t%s" % (k, path))
        for cy, cx in centres:
            labels.append("mic%d\t%d\t%d" % (k, cx, cy))
        labels += ["mic%d\t%d\t%d" % (k, 80 + 9 * j, 82 + 7 * j) for j in range(8)]   # inside the sampler's margin window
    imgs, lab = os.path.join(root, "imgs.txt"), os.path.join(root, "labels.txt")
    open(imgs, "w").write("\n".join(lines) + "\n")
    open(lab, "w").write("\n".join(labels) + "\n")
    return imgs, lab


def test_cli_train_resume_eval(tmp_path):
    from spr_pick_amd import checkpoint, cli
    from spr_pick_amd.params import ConfigValue, StateValue
    imgs, lab = _write_set(str(tmp_path))
    runs = str(tmp_path / "runs")
    argv = ("train start -a ssdn -n gaussian --noise_value var -t %s -l %s -v %s -vl %s -ap 0.75 -tau 0.01 "
            "-iter 64 --train_batch_size 16 --eval_interval 32 --print_interval 32 --checkpoint_interval 32 "
            "--nms 18 --bb 24 --runs_dir %s" % (imgs, lab, imgs, lab, runs)).split()
    trainer = cli.start(argv)
    run = os.path.join(runs, "00000-train-ssdn-gaussian-iter64-0.75-0.01-joint")
    assert trainer.run_dir_path == run and os.path.isdir(run)
    assert trainer.state[StateValue.ITERATION] == 64
    assert sorted(os.listdir(os.path.join(run, "training_jt"))) == ["model_%08d.training" % i for i in (0, 32, 64)]
    assert os.path.exists(os.path.join(run, "final-ssdn-gaussian.wt"))
    val = sorted(os.listdir(os.path.join(run, "val_imgs_joint")))
    for it in (0, 32, 64):
        for desc in ("nsy", "out", "out-mu", "out-std", "out-target", "pred_tar"):
            assert "mic0_%08d_%s.png" % (it, desc) in val
        assert "mic0_%08d_scores.txt" % it in val
    head = open(os.path.join(run, "val_imgs_joint", "mic0_00000064_scores.txt")).readline()
    assert head == "image_name\tx_coord\ty_coord\tscore\n"
    log = open(os.path.join(run, "log.txt")).read()
    assert "TRAINING STARTED" in log and "TRAINING FINISHED" in log and "[00000032] TRAIN | loss=" in log
    metrics = open(os.path.join(run, "metrics.tsv")).read()
    assert "train/loss\t32\t" in metrics and "train/learning_rate\t64\t" in metrics

    ck = checkpoint.load(os.path.join(run, "training_jt", "model_00000064.training"))
    assert sorted(ck) == ["denoiser", "optimizer", "rng", "state"]
    assert ck["state"][StateValue.ITERATION] == 64 and ck["denoiser"]["cfg"][ConfigValue.NMS] == 18
    assert all(torch.isfinite(v).all() for k, v in ck["denoiser"].items() if torch.is_tensor(v) and v.is_floating_point())
    from PIL import Image
    png = np.array(Image.open(os.path.join(run, "val_imgs_joint", "mic0_00000064_nsy.png")))
    assert png.shape == (320, 320) and png.dtype == np.uint8
    from spr_pick_amd import micrograph_io
    src = micrograph_io.load_image(os.path.join(str(tmp_path), "mic0.mrc"))
    t = src.astype(np.float32) / np.float32(255)
    lo, hi = float(t.min()), float(t.max())
    np.testing.assert_array_equal(png, np.uint8((t - lo) / (hi - lo) * 255))

    resumed = cli.start(["train", "resume", run, "--iterations", "96"])
    assert resumed.state[StateValue.ITERATION] == 96 and resumed.run_dir_path == run
    assert "model_00000096.training" in os.listdir(os.path.join(run, "training_jt"))
    w0 = ck["denoiser"]["models.denoiser_model.denoise_branch.encode_block_1.0.weight"]
    w1 = resumed.denoiser.state_dict()["models.denoiser_model.denoise_branch.encode_block_1.0.weight"].cpu()
    assert not torch.equal(w0, w1)

    ev = cli.start(["eval", "-m", os.path.join(run, "training_jt", "model_00000096.training"), "-d", imgs,
                    "--runs_dir", runs, "--nms", "18", "--num", "2"])
    out_dir = os.path.join(ev.run_dir_path, "eval_imgs")
    assert os.path.basename(ev.run_dir_path).startswith("00001-eval-ssdn-gaussian-iter96-0.75-0.01-joint")
    files = sorted(os.listdir(out_dir))
    for name in ("mic0", "mic1"):
        assert name + "_scores.txt" in files and name + "_out.png" in files and name + "_pred_tar.png" in files
    rows = open(os.path.join(out_dir, "mic1_scores.txt")).read().splitlines()[1:]
    for r in rows:
        n, x, y, s = r.split("\t")
        assert n == "mic1" and 30 < int(x) < 290 and 30 < int(y) < 290 and float(s) > 0.02
    # the weights-only file evaluates too
    ev2 = cli.start(["eval", "-m", os.path.join(run, "final-ssdn-gaussian.wt"), "-d", imgs, "--runs_dir", runs,
                     "--num", "1"])
    assert os.path.exists(os.path.join(ev2.run_dir_path, "eval_imgs", "mic0_scores.txt"))


def test_denoise_only_mode(tmp_path):
    from spr_pick_amd import cli
    from spr_pick_amd.params import StateValue
    imgs, lab = _write_set(str(tmp_path), n=1)
    runs = str(tmp_path / "runs")
    t = cli.start(("train start -a ssdn -n gaussian --noise_value var --dn_only -t %s -l %s -ap 0.75 -tau 0.01 "
                   "-iter 32 --train_batch_size 16 --print_interval 16 --checkpoint_interval 32 --runs_dir %s"
                   % (imgs, lab, runs)).split())
    assert t.state[StateValue.ITERATION] == 32 and t.mode == "denoise"
    assert os.path.isdir(os.path.join(t.run_dir_path, "training_dn"))


def test_two_rank_training_and_sharded_eval(tmp_path):
    """`joint train start` under torch.distributed.run with 2 ranks (gloo rehearsal of the RCCL path on
    one GPU): one run directory, rank-0 checkpoints, global-batch iteration counting; then a 2-rank
    eval where micrograph i is written by rank i % 2."""
    import socket
    import subprocess
    import sys
    from spr_pick_amd import checkpoint
    from spr_pick_amd.params import StateValue
    imgs, lab = _write_set(str(tmp_path))
    runs = str(tmp_path / "runs")
    env = dict(os.environ, SPRK_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

    def launch(args):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "spr_pick_amd", "--"] + args
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]

    launch(("train start -a ssdn -n gaussian --noise_value var -t %s -l %s -ap 0.75 -tau 0.01 -iter 64 "
            "--train_batch_size 16 --print_interval 32 --checkpoint_interval 64 --runs_dir %s" % (imgs, lab, runs)).split())
    assert os.listdir(runs) == ["00000-train-ssdn-gaussian-iter64-0.75-0.01-joint"]
    run = os.path.join(runs, os.listdir(runs)[0])
    assert sorted(os.listdir(os.path.join(run, "training_jt"))) == ["model_00000000.training", "model_00000064.training"]
    ck = checkpoint.load(os.path.join(run, "training_jt", "model_00000064.training"))
    assert ck["state"][StateValue.ITERATION] == 64
    # 64 images at a global batch of 16 = 4 optimiser steps on every rank
    assert int(next(iter(ck["optimizer"]["state"].values()))["step"]) == 4
    launch(["eval", "-m", os.path.join(run, "final-ssdn-gaussian.wt"), "-d", imgs, "--runs_dir", runs, "--num", "2"])
    ev = [d for d in os.listdir(runs) if "-eval-" in d]
    assert len(ev) == 1
    files = os.listdir(os.path.join(runs, ev[0], "eval_imgs"))
    assert "mic0_scores.txt" in files and "mic1_scores.txt" in files


def test_eval_reports_psnr_against_reference_images(tmp_path):
    """`joint eval --gt_dataset`: psnr_out / psnr_mu_out = -10 log10(mean squared error) of the un-padded
    outputs against the clean images (train.py:404-413, :781-814)."""
    from spr_pick_amd import cli, micrograph_io
    from spr_pick_amd.params import HistoryValue, StateValue
    imgs, lab = _write_set(str(tmp_path), n=1)
    runs = str(tmp_path / "runs")
    t = cli.start(("train start -a ssdn -n gaussian --noise_value var -t %s -l %s -ap 0.75 -tau 0.01 -iter 16 "
                   "--train_batch_size 16 --print_interval 16 --checkpoint_interval 16 --runs_dir %s"
                   % (imgs, lab, runs)).split())
    # the "clean" set: the same micrograph (any image of the right size would do)
    gt_dir = tmp_path / "gt"
    gt_dir.mkdir()
    src = micrograph_io.load_image(os.path.join(str(tmp_path), "mic0.mrc"))
    with open(gt_dir / "mic0.mrc", "wb") as f:
        micrograph_io.write_mrc(f, src.astype(np.float32))
    ev = cli.start(["eval", "-m", os.path.join(t.run_dir_path, "final-ssdn-gaussian.wt"), "-d", imgs, "-g", str(gt_dir),
                    "--runs_dir", runs, "--num", "1"])
    h = ev.state[StateValue.HISTORY][HistoryValue.EVAL]
    assert h["n"] == 1 and not h["psnr_out"].empty() and not h["psnr_mu_out"].empty()
    psnr = float(h["psnr_out"].accumulated())
    from PIL import Image
    out = np.array(Image.open(os.path.join(ev.run_dir_path, "eval_imgs", "mic0_out.png")))
    assert out.shape == (320, 320) and np.isfinite(psnr) and 0.0 < psnr < 80.0
    log = open(os.path.join(ev.run_dir_path, "log.txt")).read()
    assert "psnr_out=" in log and "psnr_mu_out=" in log


def test_rccl_at_world_size_one_bench_and_trainer(tmp_path):
    """RCCL has to meet this code before an 8-GPU box does (north_star: "RCCL all-reduce of gradients over xGMI").  With
    SPRK_DIST_FORCE=1 a one-rank process group is created on backend "nccl" (= RCCL on ROCm) and the in-place
    all-reduce of the flat gradient buffer runs every step: RCCL initialisation, torch.cuda.set_device ordering, the
    HIP-graph capture in "thread_local" mode beside RCCL's watchdog thread, and replay + collective + Adam in a loop —
    for bench.py (which also checks replay == eager in-process) and for the `joint train start` CLI."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def env():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        e = dict(os.environ, SPRK_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)
        e.pop("SPRK_DIST_BACKEND", None)
        return e

    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--infer-size", "0",
                        "--infer-large", "0", "--no-cpu-baseline", "--also-dtype", "none", "--sustain-seconds", "0",
                        "--batch16", "off", "--event-steps", "1"], env=env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert "RCCL" in out["config"]["parallelism"] and "forced 1-rank" in out["config"]["parallelism"], out["config"]
    assert "HIP graphs" in out["config"]["execution"] and out["graph_fallback"] is None
    assert out["bit_identical"] is True and np.isfinite(out["final_loss"]) and out["value"] > 100

    imgs, lab = _write_set(str(tmp_path))
    runs = str(tmp_path / "runs")
    r = subprocess.run([sys.executable, "-m", "spr_pick_amd"] +
                       ("train start -a ssdn -n gaussian --noise_value var -t %s -l %s -ap 0.75 -tau 0.01 -iter 96 "
                        "--train_batch_size 16 --print_interval 32 --checkpoint_interval 96 --runs_dir %s"
                        % (imgs, lab, runs)).split(), env=env(), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    run = os.path.join(runs, os.listdir(runs)[0])
    log = open(os.path.join(run, "log.txt")).read()
    assert "HIP-graph replay" in log and "gradient collectives: 6 over RCCL, world size 1" in log, log[-1500:]
