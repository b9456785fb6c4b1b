"""GPU parity of the HIP-backed networks / Denoiser against the oracle and the golden vectors
generated from the reference (tests/golden), plus NMS coordinate parity (bit-exact)."""
import numpy as np
import pytest
import torch

from conftest import check_probe, golden

pytestmark = pytest.mark.gpu

# End-to-end fp32 tolerance.  ~30 chained convolutions, each a differently-ordered fp32 sum than
# the CPU reference: 1e-4 relative to each tensor's max |value| (observed ~1e-5).
REL = 1e-4


def close(got, want, rel=REL, name=""):
    got = np.asarray(got.detach().cpu() if torch.is_tensor(got) else got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = np.abs(want).max() + 1e-30
    worst = np.abs(got - want).max()
    assert worst <= rel * scale, "%s: max err %.3e vs scale %.3e (rel %.2e)" % (name, worst, scale, worst / scale)


def make_cfg():
    from spr_pick_amd import cfg, params
    c = cfg.base()
    c[params.ConfigValue.ALGORITHM] = params.NoiseAlgorithm.SELFSUPERVISED_DENOISING
    c[params.ConfigValue.NOISE_STYLE] = "gaussian"
    c[params.ConfigValue.NOISE_VALUE] = params.NoiseValue.UNKNOWN_VARIABLE
    c[params.ConfigValue.NMS] = 18
    return cfg.infer(c, model_only=True)


@pytest.fixture()
def denoiser(oracle_state):
    from spr_pick_amd import Denoiser
    den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
    sd = {"models." + k: v for k, v in oracle_state.items()}
    missing, unexpected = den.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("_models.") for m in missing)
    return den


def test_state_dict_layout_matches_reference(denoiser):
    import json, os
    from conftest import GOLDEN
    layout = json.load(open(os.path.join(GOLDEN, "state_layout.json")))
    ours = denoiser.state_dict()
    assert set(ours) == set(layout)
    for k, v in ours.items():
        if k != "cfg":
            assert list(v.shape) == layout[k], k


def test_unet_forward_and_blindspot(denoiser):
    g = golden("unet_fwd.npz")
    net = denoiser.models["denoiser_model"].denoise_branch
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        out, _ = net(x)
        close(out, g["out_stats"], name="out_stats")
        x2 = x.clone()
        x2[0, 0, 20, 37] += 0.25
        out2, _ = net(x2)
    # blind-spot property, bit-exact: the perturbed pixel's own output cannot move
    assert torch.equal(out[0, :, 20, 37], out2[0, :, 20, 37])
    assert not torch.equal(out[0], out2[0])


def test_sigma_net_and_detector_modes(denoiser):
    g = golden("parts.npz")
    jn = denoiser.models["denoiser_model"]
    sg = denoiser.models["sigma_estimation_model"]
    with torch.no_grad():
        close(sg(torch.from_numpy(g["x"]).cuda()), g["sigma_out"], name="sigma")
        z, zf = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(g["zf"]).cuda()
        det = jn.detector
        det.eval(); det.unfill()
        close(det(z), g["det_eval_unfilled"], name="det eval unfilled")
        assert det.fill() == 4
        close(det(zf), g["det_eval_filled"], name="det eval filled")
        close(det(z[:1]), g["det_eval_filled64"], name="det eval filled 64")
        det.unfill()
    det.train()
    out = det(z)
    close(out, g["det_train_unfilled"], name="det train")
    sd = det.state_dict()
    for k in g.files:
        if k.startswith("bn_after/"):
            name = k[len("bn_after/"):]
            close(sd[name].float(), g[k].astype(np.float64), rel=1e-5, name=name)


@pytest.mark.parametrize("tag", ["w", "h"])
def test_joint_train_step_matches_reference(denoiser, tag):
    from spr_pick_amd import DetectionDataset
    from spr_pick_amd.params import PipelineOutput as P
    g = golden("joint_train_%s.npz" % tag)
    denoiser.train(); denoiser.unfill()
    data = DetectionDataset.make_batch(torch.from_numpy(g["inp"]), torch.from_numpy(g["target"]))
    o = denoiser.run_pipeline(data, float(g["alpha"]), float(g["tau"]), train=True,
                              eps=torch.from_numpy(g["eps"]).cuda(), eps_flip=torch.from_numpy(g["eps_flip"]).cuda(),
                              flip_p=float(g["flip_p"]))
    torch.mean(o[P.LOSS]).backward()
    for key in ("LOSS", "DENOISE_LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        close(o[getattr(P, key)], g[key], name=key)
    close(o[P.DETECT_LOSS].reshape(()), g["DETECT_LOSS"], name="DETECT_LOSS")
    close(o[P.AUG_LOSS].reshape(()), g["AUG_LOSS"], rel=1e-3, name="AUG_LOSS")
    nograd = set(g["nograd"].tolist())
    out_of_budget = []
    for name, p in denoiser.models.named_parameters():
        if name in nograd:
            assert p.grad is None, name
        else:
            assert p.grad is not None, name
            key = "grad/" + name
            a = p.grad.detach().cpu().numpy().astype(np.float64).ravel()
            absmax = float(g[key + "/absmax"])
            err = np.abs(a[g[key + "/idx"]] - g[key + "/val"]).max()
            # Gradient tolerance: 3e-3 of the tensor's max |g| (plus 1e-4 absolute for the two
            # BatchNorm-before-BatchNorm parameters whose true gradient is zero).  It is set by the
            # network, not by the kernels: a (Leaky)ReLU pre-activation within fp32 rounding of 0
            # takes slope 1 on one implementation and 0.1 on another.  Measured on this fixture
            # against an fp64 evaluation: exactly 2 of the 96 channels of output_block.2.bias differ
            # (by 7.6e-4 of max|g|), all others by 3e-6; the direct (non-MFMA) kernels show the
            # same two flips.  Per-operator gradients are checked at 5e-5 in test_gpu_ops.py.
            assert err <= 3e-3 * absmax + 1e-4, "%s: probe err %.3e vs max|g| %.3e" % (name, err, absmax)
            assert abs(np.linalg.norm(a) / float(g[key + "/norm"]) - 1) < 2e-3 or absmax < 1e-3, name
            # ... and the 3e-3 is for the sign-flip casualties only: over all tensors at most 0.5 % of the sampled
            # gradient values may be beyond 1e-3 of their tensor's max|g| (counted below: 0.1 %)
            errs = np.abs(a[g[key + "/idx"]] - g[key + "/val"])
            loose = int((errs > 1e-3 * absmax + 1e-6).sum())
            out_of_budget.append((name, loose, len(errs), float(errs.max() / (absmax + 1e-30))))
    total_loose, total = sum(n for _, n, _, _ in out_of_budget), sum(m for _, _, m, _ in out_of_budget)
    worst = sorted(out_of_budget, key=lambda t: -t[3])[:4]
    print("gradient probes beyond 1e-3 of max|g|: %d of %d; worst tensors %s" % (
        total_loose, total, [(n.split("denoiser_model.")[-1], "%.1e" % w) for n, _, _, w in worst]))
    # measured: 2-4 of 14 470 (the two BatchNorm-before-BatchNorm parameters whose true gradient is zero)
    assert total_loose <= total // 1000, "more than 0.1 %% of the gradient probes are beyond 1e-3 of max|g|"
    sd = denoiser.models["denoiser_model"].detector.state_dict()
    for k in g.files:
        if k.startswith("bn_after/"):
            name = k[len("bn_after/"):]
            close(sd[name].float(), g[k].astype(np.float64), rel=1e-4, name=name)


def test_poisson_branch_matches_reference(oracle_state):
    """--noise_style poisson (denoiser_v2.py:412-424) against the reference's own outputs (tests/golden/joint_poisson.npz,
    generator oracle/gen_golden_poisson.py): a joint train step (outputs, losses, parameter gradients) and a filled eval
    pass; NOISE_STD_DEV is a per-pixel map in this branch."""
    from spr_pick_amd import Denoiser, DetectionDataset
    from spr_pick_amd.params import ConfigValue, PipelineOutput as P
    g = golden("joint_poisson.npz")
    cfg = make_cfg()
    cfg[ConfigValue.NOISE_STYLE] = "poisson"
    den = Denoiser(cfg, device="cuda:0", mode="joint")
    sd = {"models." + k: v.clone() for k, v in oracle_state.items()}
    sd["models.denoiser_model.denoise_branch.output_conv.bias"][0] += float(g["mu_bias"])
    _, unexpected = den.load_state_dict(sd, strict=False)
    assert not unexpected
    den.train(); den.unfill()
    data = DetectionDataset.make_batch(torch.from_numpy(g["inp"]), torch.from_numpy(g["target"]))
    o = den.run_pipeline(data, float(g["alpha"]), float(g["tau"]), train=True, eps=torch.from_numpy(g["eps"]).cuda(),
                         eps_flip=torch.from_numpy(g["eps_flip"]).cuda(), flip_p=float(g["flip_p"]))
    torch.mean(o[P.LOSS]).backward()
    assert tuple(o[P.NOISE_STD_DEV].shape) == (3, 64, 64)
    # (the posterior mean is a ratio of two variances that are both ~1e-5 where mu sits on the floor: 5e-4, measured 2e-4)
    for key in ("LOSS", "DENOISE_LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        close(o[getattr(P, key)], g[key], name=key, rel=5e-4 if key == "IMG_DENOISED" else REL)
    close(o[P.DETECT_LOSS].reshape(()), g["DETECT_LOSS"], name="DETECT_LOSS")
    loose = total = 0
    for name, p in den.models.named_parameters():
        key = "grad/" + name
        if key + "/val" not in g.files:
            assert p.grad is None, name
            continue
        a = p.grad.detach().cpu().numpy().astype(np.float64).ravel()
        absmax = float(g[key + "/absmax"])
        errs = np.abs(a[g[key + "/idx"]] - g[key + "/val"])
        assert errs.max() <= 3e-3 * absmax + 1e-4, "%s: probe err %.3e vs max|g| %.3e" % (name, errs.max(), absmax)
        loose += int((errs > 1e-3 * absmax + 1e-6).sum())
        total += len(errs)
    assert total > 10000 and loose <= total // 500, (loose, total)
    den.load_state_dict(sd, strict=False)
    den.eval(); den.fill()
    with torch.no_grad():
        oe = den.run_pipeline(DetectionDataset.make_batch(torch.from_numpy(g["eval/inp"]), torch.zeros(1, 1)), train=False,
                              eps=torch.from_numpy(g["eval/eps"]).cuda())
    den.unfill()
    for key in ("LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        close(oe[getattr(P, key)], g["eval/" + key], name="eval " + key, rel=5e-4 if key == "IMG_DENOISED" else REL)


def test_joint_eval_and_picks(denoiser, oracle_state):
    from oracle import nms, pipeline
    from spr_pick_amd import DetectionDataset, non_maximum_suppression
    from spr_pick_amd.params import PipelineOutput as P
    g = golden("joint_eval.npz")
    denoiser.eval(); denoiser.fill()
    with torch.no_grad():
        o = denoiser.run_pipeline(DetectionDataset.make_batch(torch.from_numpy(g["inp"]), torch.zeros(1, 1)),
                                  train=False, eps=torch.from_numpy(g["eps"]).cuda())
    denoiser.unfill()
    for key in ("LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        close(o[getattr(P, key)], g[key], name=key)
    # coordinate parity, bit-exact: (1) HIP NMS on the reference's score map == reference picks
    for r in (18, 5):
        s, c = non_maximum_suppression(g["DETECT"][0, 0], r, set(), 0.02)
        assert np.array_equal(c, g["nms%d_coords" % r]) and np.array_equal(s, g["nms%d_scores" % r])
    # (2) end to end: HIP scores -> HIP NMS == oracle NMS on the same HIP scores
    score = o[P.DETECT][0, 0]
    s, c = non_maximum_suppression(score, 18, set(), 0.02)
    s2, c2 = nms.nms_c(score.cpu().numpy(), 18, 0.02)
    assert np.array_equal(c, c2) and np.array_equal(s, s2)


def test_ssdn_pipeline(oracle_state):
    from spr_pick_amd import Denoiser, DetectionDataset
    from spr_pick_amd.params import PipelineOutput as P
    g = golden("ssdn_eval.npz")
    den = Denoiser(make_cfg(), device="cuda:0", mode="denoise")
    den.load_state_dict({"models." + k: v for k, v in oracle_state.items()}, strict=False)
    den.eval()
    with torch.no_grad():
        o = den.run_pipeline(DetectionDataset.make_batch(torch.from_numpy(g["inp"]), torch.zeros(2, 1)))
    for key in ("LOSS", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        close(o[getattr(P, key)], g[key], name=key)


def nms_case_names():
    g = golden("nms_cases.npz")
    return sorted({k.split("/")[0] for k in g.files})


@pytest.mark.parametrize("name", nms_case_names())
def test_nms_golden_cases(name):
    from spr_pick_amd import non_maximum_suppression
    g = golden("nms_cases.npz")
    s, c = non_maximum_suppression(g[name + "/x"], int(g[name + "/r"]), set(), float(g[name + "/thr"]))
    assert c.dtype == np.int32 and s.dtype == np.float32 and c.shape == (len(s), 2)
    assert np.array_equal(c, g[name + "/coords"].reshape(-1, 2)), name
    assert np.array_equal(s, g[name + "/scores"]), name


@pytest.mark.parametrize("shape,r,thr,kind", [
    ((512, 512), 18, 0.02, "rand"), ((300, 700), 7, 0.5, "rand"), ((1024, 1024), 18, 0.02, "blobs"),
    ((257, 129), 40, 0.1, "rand"), ((128, 128), 3, 0.02, "ties"), ((400, 400), 18, 0.02, "ramp"),
])
def test_nms_vs_oracle_large(shape, r, thr, kind):
    """Random / peaky / tied / monotone-ramp maps (long dependency chains) against the C oracle."""
    from oracle import nms
    from spr_pick_amd import non_maximum_suppression
    rng = np.random.default_rng(7)
    H, W = shape
    if kind == "rand":
        x = rng.random(shape, dtype=np.float32)
    elif kind == "ties":
        x = rng.integers(0, 6, size=shape).astype(np.float32) / 5.0
    elif kind == "ramp":
        yy, xx = np.mgrid[0:H, 0:W]
        x = ((yy * 3 + xx * 2) / (5.0 * max(H, W))).astype(np.float32)
    else:
        yy, xx = np.mgrid[0:H, 0:W]
        x = rng.random(shape) * 0.015
        for _ in range(300):
            cy, cx, a = rng.integers(0, H), rng.integers(0, W), rng.random() * 0.8 + 0.1
            x += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 32.0)
        x = np.clip(x, 0, 0.9999).astype(np.float32)
    s, c = non_maximum_suppression(x, r, set(), thr)
    s2, c2 = nms.nms_c(x, r, thr)
    assert len(s) == len(s2)
    assert np.array_equal(c, c2) and np.array_equal(s, s2)


def test_nms_full_size_properties():
    """4096x4096 (BASELINE config 3 size): size-independent properties instead of the oracle."""
    from spr_pick_amd import nms_device
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.rand((4096, 4096), generator=g, device="cuda") ** 8
    r, thr = 18, 0.02
    s, c = nms_device(x, r, thr)
    s, c = s.cpu().numpy(), c.cpu().numpy().astype(np.int64)
    assert len(s) > 1000 and np.all(s > thr)
    assert np.all(np.diff(s) <= 0)                                   # sorted by descending score
    assert np.array_equal(x.cpu().numpy()[c[:, 1], c[:, 0]], s)      # scores are the map's values
    # picks are pairwise farther than r apart (checked on a grid hash)
    cell = {}
    for i, (px, py) in enumerate(c):
        cell.setdefault((px // 19, py // 19), []).append(i)
    for (gx, gy), idx in cell.items():
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for j in cell.get((gx + dx, gy + dy), []):
                    for i in idx:
                        if i < j:
                            d2 = (c[i, 0] - c[j, 0]) ** 2 + (c[i, 1] - c[j, 1]) ** 2
                            assert d2 > r * r
    # idempotence: suppressing the picked-only map returns the same picks
    y = torch.zeros_like(x)
    y[torch.from_numpy(c[:, 1]).cuda(), torch.from_numpy(c[:, 0]).cuda()] = torch.from_numpy(s).cuda()
    s3, c3 = nms_device(y, r, thr)
    assert np.array_equal(c3.cpu().numpy(), c) and np.array_equal(s3.cpu().numpy(), s)


def test_training_reduces_loss_and_is_deterministic(oracle_state):
    """Three optimiser steps on a fixed batch: finite, decreasing loss, and two identical runs
    produce bit-identical parameters (all reductions are order-fixed)."""
    from spr_pick_amd import Denoiser, DetectionDataset
    from spr_pick_amd.params import PipelineOutput as P
    g = golden("joint_train_w.npz")

    def run():
        den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
        den.load_state_dict({"models." + k: v for k, v in oracle_state.items()}, strict=False)
        den.train()
        opt = torch.optim.Adam([p for p in den.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.99))
        losses = []
        for _ in range(3):
            opt.zero_grad()
            o = den.run_pipeline(DetectionDataset.make_batch(torch.from_numpy(g["inp"]), torch.from_numpy(g["target"])),
                                 0.75, 0.01, train=True, eps=torch.from_numpy(g["eps"]).cuda(),
                                 eps_flip=torch.from_numpy(g["eps_flip"]).cuda(), flip_p=0.3)
            loss = torch.mean(o[P.LOSS])
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses, torch.cat([p.detach().reshape(-1) for p in den.models.parameters()])

    l1, p1 = run()
    l2, p2 = run()
    assert all(np.isfinite(l1)) and l1[-1] < l1[0]
    assert l1 == l2 and torch.equal(p1, p2)


# ---- size-independent properties at BASELINE's full sizes ------------------------------
def test_full_batch_directional_derivative(oracle_state):
    """configs[1] (batch 32, 64x64 patches): the gradient the backward kernels produce predicts the change of
    the loss along a random parameter direction,  (L(w + h d) - L(w - h d)) / 2h  ~=  <grad, d>.
    The fixed eps / flip draws make L a deterministic function of w."""
    from spr_pick_amd import Denoiser, DetectionDataset, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
    den.load_state_dict({"models." + k: v for k, v in oracle_state.items()}, strict=False)
    den.train()
    mics = [synthetic.micrograph(i) for i in range(4)]
    inp, tgt = synthetic.patch_batches(1, 32, mics, device="cuda:0")[0]
    gen = torch.Generator(device="cuda").manual_seed(5)
    eps = torch.randn(inp.shape, device="cuda", generator=gen)
    eps_f = torch.randn(inp.shape, device="cuda", generator=gen)
    # BatchNorm running statistics change on every training pass; they do not enter the training-mode loss
    params = [p for n, p in den.named_parameters() if n.startswith("models.") and p.requires_grad]

    def loss():
        o = den.run_pipeline(DetectionDataset.make_batch(inp, tgt), 0.75, 0.01, train=True, eps=eps, eps_flip=eps_f,
                             flip_p=0.3)
        return torch.mean(o[P.LOSS])

    L0 = loss()
    grads = torch.autograd.grad(L0, params, allow_unused=True)
    live = [(p, g) for p, g in zip(params, grads) if g is not None]
    assert len(live) >= 90   # 12 of the parameter tensors never receive a gradient
    dirs = [torch.randn(p.shape, device="cuda", generator=gen) * p.detach().abs().mean().clamp_min(1e-3)
            for p, _ in live]
    pred = float(sum((g.double() * d.double()).sum() for (_, g), d in zip(live, dirs)))
    vals = {}
    for h in (2e-3, 1e-3):
        with torch.no_grad():
            for (p, _), d in zip(live, dirs):
                p.add_(d, alpha=h)
            lp = float(loss())
            for (p, _), d in zip(live, dirs):
                p.add_(d, alpha=-2 * h)
            lm = float(loss())
            for (p, _), d in zip(live, dirs):
                p.add_(d, alpha=h)
        vals[h] = (lp - lm) / (2 * h)
    # central differences in fp32 over ~50 layers: a few % (LeakyReLU kinks, rounding of L ~ 40 at h ~ 1e-3)
    assert abs(vals[1e-3] - pred) <= 0.05 * abs(pred) + 1e-3, (pred, vals)
    assert abs(vals[2e-3] - pred) <= 0.08 * abs(pred) + 1e-3, (pred, vals)


def test_filled_inference_is_translation_consistent(oracle_state):
    """configs[2]-style whole-micrograph inference: the filled network is a stack of convolutions and 2x2
    poolings, so the score map and the denoised image of a crop (offsets a multiple of 32 = 2^5 pooling
    levels) agree with the full micrograph's away from the crop's border — whatever tiles, chunks and
    workgroup shapes the planner picked for the two sizes."""
    from spr_pick_amd import Denoiser, DetectionDataset, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
    den.load_state_dict({"models." + k: v for k, v in oracle_state.items()}, strict=False)
    den.eval()
    den.fill()
    S, C, off = 2048, 1024, 512
    img = torch.from_numpy(synthetic.micrograph(3, size=S)[0].astype(np.float32) / 255.0).cuda()[None, None]
    crop = img[:, :, off:off + C, off:off + C].contiguous()
    with torch.no_grad():
        full = den.run_pipeline(DetectionDataset.make_batch(img, torch.zeros(1, 1)), train=False,
                                eps=torch.zeros_like(img))
        part = den.run_pipeline(DetectionDataset.make_batch(crop, torch.zeros(1, 1)), train=False,
                                eps=torch.zeros_like(crop))
    m = 448   # beyond the receptive field of the 5-level blind-spot U-Net + detector (pad 31)
    for key in (P.IMG_MU, P.DETECT):
        a = full[key][0, 0, off + m:off + C - m, off + m:off + C - m]
        b = part[key][0, 0, m:C - m, m:C - m]
        scale = float(a.abs().max())
        worst = float((a - b).abs().max())
        print(key, "max |diff| %.3e of scale %.3e; bitwise equal: %s" % (worst, scale, bool(torch.equal(a, b))))
        assert worst <= 2e-5 * scale + 1e-7, key
    den.unfill()


def test_filled_inference_4096_and_nms(oracle_state):
    """BASELINE configs[2] at full size: one 4096x4096 micrograph through the filled forward (sigma net,
    posterior mean, clamped sigmoid) and the NMS.  4096^2 crosses the 2^31-element / 2 GB buffer-range limits the
    kernels special-case, so nothing smaller exercises those paths.  Checks:
      * three 1024^2 interior windows (top-left, centre, bottom-right quadrant) against the SAME network on the
        cropped input (translation consistency, as above) at 2e-5 of the window's max |value|;
      * the picks of the device NMS on the full 4096^2 score map against the C oracle on that same map,
        coordinates and scores bit-exact (reference: utils/algorithms.py:59-103, call train.py:564, r = 18,
        threshold 0.02);
      * the reference's border post-filter (train.py:563-571) keeps only interior picks."""
    from oracle import nms
    from spr_pick_amd import Denoiser, DetectionDataset, nms_device, picks, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
    den.load_state_dict({"models." + k: v for k, v in oracle_state.items()}, strict=False)
    den.eval()
    den.fill()
    S, C = 4096, 1024
    img = torch.from_numpy(synthetic.micrograph(5, size=S)[0].astype(np.float32) / 255.0).cuda()[None, None]
    with torch.no_grad():
        full = den.run_pipeline(DetectionDataset.make_batch(img, torch.zeros(1, 1)), train=False,
                                eps=torch.zeros_like(img))
        mu, det = full[P.IMG_MU][0, 0].clone(), full[P.DETECT][0, 0].clone()
        del full
        torch.cuda.empty_cache()
        m = 448
        for off_y, off_x in ((0, 0), (1536, 1536), (3072, 2048)):
            crop = img[:, :, off_y:off_y + C, off_x:off_x + C].contiguous()
            part = den.run_pipeline(DetectionDataset.make_batch(crop, torch.zeros(1, 1)), train=False,
                                    eps=torch.zeros_like(crop))
            for name, whole, key in (("mu", mu, P.IMG_MU), ("detect", det, P.DETECT)):
                a = whole[off_y + m:off_y + C - m, off_x + m:off_x + C - m]
                b = part[key][0, 0, m:C - m, m:C - m]
                scale, worst = float(a.abs().max()), float((a - b).abs().max())
                assert worst <= 2e-5 * scale + 1e-7, (name, off_y, off_x, worst, scale)
    den.unfill()
    s, c = nms_device(det, 18, 0.02)
    s, c = s.cpu().numpy(), c.cpu().numpy()
    s2, c2 = nms.nms_c(det.cpu().numpy(), 18, 0.02)
    assert len(s) == len(s2) and len(s) > 1000, (len(s), len(s2))
    assert np.array_equal(c, c2) and np.array_equal(s, s2)
    keep = picks.filter_picks(s, c, (S, S))
    kc = np.asarray(keep[1])
    assert len(kc) and kc.min() > 30 and kc.max() < S - 30


# network-level budgets of the 16-bit-operand path (see the test's docstring): relative RMS of outputs / losses,
# minimum cosine and norm deviation of every parameter gradient against the exact model's
#   out = 4 u sqrt(L): u = 2^-8 / 2^-11, L = 23 convolutions on the U-Net's longest path (independent roundings add
#   in quadrature: 2 u sqrt(L) at the U-Net output, doubled for the detector / variance division behind it);
#   measured on the golden step: bf16 5.2 % (DETECT) and <= 1.8 % elsewhere, fp16 0.61 % and <= 0.23 %.
#   Gradients, against the exact model's (the network is chaotic in the roundings — a one-ulp fp32 difference that
#   flips a 16-bit rounding grows about 3x per layer — so this is a statistical statement): measured minimum cosine
#   0.846 (bf16) / 0.990 (fp16), norms within 39 % / a few %.
BUDGET16 = {"bf16": {"out": 4 * 2.0 ** -8 * 23 ** 0.5, "cos": 0.75, "norm": 0.6},
            "f16": {"out": 4 * 2.0 ** -11 * 23 ** 0.5, "cos": 0.97, "norm": 0.15}}


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_joint_train_step_16bit_operands(denoiser, oracle_state, dt):
    """BASELINE configs[4]: the joint training step with the U-Nets' MFMA operands in bf16 / fp16 (tensors, master
    weights, accumulation and every non-convolution kernel fp32).

    Kernel-level exactness is pinned per operator (test_gpu_ops.py: every output equals conv(round16(x), round16(w))
    with exact products and fp32 sums to 2e-5).  At network level two statements, both statistical (relative RMS;
    budgets and measured values at BUDGET16 above):
    (1) against the oracle evaluated with the same EXACT MODEL (oracle/networks.py MODEL16: operands of the eligible
        U-Net convolutions rounded in forward and backward-data, fp32 backward-weight) — also checks that the library
        takes the 16-bit kernels for exactly the layers the model rounds (46 launches);
    (2) against the reference's fp32 golden step: the precision a user trades for the speed.
    Forced 16-bit ("bf16!") so that every layer the kernels cover is exercised, not only those where they are faster."""
    import copy
    from oracle import networks as onet
    from oracle import pipeline
    from spr_pick_amd import DetectionDataset, _lib
    from spr_pick_amd.params import PipelineOutput as P
    g = golden("joint_train_w.npz")
    denoiser.train(); denoiser.unfill()
    n_layers = denoiser.set_conv_dtype(dt + "!")
    assert n_layers == 19 + 17       # every conv of the two U-Nets except their final output convolutions
    L = _lib.lib()
    n0 = L.sprk_conv16_launch_count()
    data = DetectionDataset.make_batch(torch.from_numpy(g["inp"]), torch.from_numpy(g["target"]))
    try:
        o = denoiser.run_pipeline(data, float(g["alpha"]), float(g["tau"]), train=True,
                                  eps=torch.from_numpy(g["eps"]).cuda(), eps_flip=torch.from_numpy(g["eps_flip"]).cuda(),
                                  flip_p=float(g["flip_p"]))
        torch.mean(o[P.LOSS]).backward()
    finally:
        denoiser.set_conv_dtype("f32")
    ran16 = L.sprk_conv16_launch_count() - n0
    assert ran16 >= 40, "only %d convolution launches took the 16-bit kernels" % ran16

    sd = {k: v.clone() for k, v in oracle_state.items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    onet.MODEL16["dtype"] = torch.bfloat16 if dt == "bf16" else torch.float16
    try:
        res = pipeline.joint_pipeline(sd, torch.from_numpy(g["inp"]), torch.from_numpy(g["target"]), float(g["alpha"]),
                                      float(g["tau"]), True, torch.from_numpy(g["eps"]), torch.from_numpy(g["eps_flip"]),
                                      float(g["flip_p"]))
        res["LOSS"].mean().backward()
    finally:
        onet.MODEL16["dtype"] = None
    def rel_rms(got, want):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        return float(np.sqrt(np.mean((got - want) ** 2)) / (np.sqrt(np.mean(want ** 2)) + 1e-30))
    model_err, ref_err = {}, {}
    for key in ("LOSS", "DENOISE_LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        got = o[getattr(P, key)].detach().cpu().numpy()
        model_err[key] = rel_rms(got, res[key].detach().numpy())
        ref_err[key] = rel_rms(got, g[key])
    cos_min, norm_dev = 1.0, 0.0
    biggest = max(float(v.grad.double().norm()) for v in sd.values() if v.requires_grad and v.grad is not None)
    for name, p in denoiser.models.named_parameters():
        want = sd[name].grad
        if p.grad is None:
            assert want is None, name
            continue
        a, b = p.grad.detach().cpu().double().ravel(), want.double().ravel()
        if float(b.norm()) < 1e-2 * biggest:     # (near-)zero gradients, e.g. BatchNorm before BatchNorm: pure noise
            continue
        cos_min = min(cos_min, float(a @ b / (a.norm() * b.norm() + 1e-30)))
        norm_dev = max(norm_dev, abs(float(a.norm() / b.norm()) - 1.0))
    print("%s: %d 16-bit launches; rel. RMS vs exact model %s; vs fp32 reference %s; gradients vs exact model: min cosine "
          "%.5f, worst |norm ratio - 1| %.4f" % (dt, ran16, {k: "%.1e" % v for k, v in model_err.items()},
                                                  {k: "%.1e" % v for k, v in ref_err.items()}, cos_min, norm_dev))
    budget = BUDGET16[dt]
    assert all(v <= budget["out"] for v in model_err.values()), model_err
    assert all(v <= budget["out"] for v in ref_err.values()), ref_err
    assert cos_min >= budget["cos"] and norm_dev <= budget["norm"], (cos_min, norm_dev)


def test_nms_with_seeded_contam_set():
    """The reference's `contam` argument (utils/algorithms.py:77,98-101): pre-suppressed indices are skipped, the set is
    extended by everything the picks suppress — against the literal oracle walk, picks and final set identical."""
    from oracle import nms
    from spr_pick_amd import non_maximum_suppression
    rng = np.random.default_rng(5)
    x = rng.random((96, 80), dtype=np.float32)
    seed = set(int(v) for v in rng.integers(0, 96 * 80, size=400)) | {96 * 80 + 3}     # incl. one past-the-end index
    mine, ref = set(seed), set(seed)
    s, c = non_maximum_suppression(x, 6, mine, 0.3)
    s2, c2 = nms.nms_literal(x, 6, 0.3, contam=ref)
    assert len(s) > 10 and np.array_equal(c, c2) and np.array_equal(s, s2)
    assert mine == ref
    flat = c[:, 1].astype(np.int64) * 80 + c[:, 0]
    assert not (set(flat.tolist()) & seed)
    empty = set()
    non_maximum_suppression(x, 6, empty, 0.3)
    assert empty == set()      # an empty set is left alone (documented; algorithms.UPDATE_EMPTY_CONTAM)


def _filled_denoiser(oracle_state):
    from spr_pick_amd import Denoiser
    den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
    den.load_state_dict({"models." + k: v for k, v in oracle_state.items()}, strict=False)
    den.eval()
    den.fill()
    return den


EVAL_KEYS = ("IMG_MU", "IMG_DENOISED", "DETECT", "MODEL_STD_DEV", "NOISE_STD_DEV", "LOSS")


def test_halo_tiled_inference_is_bit_identical_to_whole_image(oracle_state):
    """SURVEY §7 step 6 (J3).  Halo-tiled evaluation of the filled pipeline (Denoiser._tiled_networks) on a 3072^2
    micrograph, windows of 2048^2: every output BIT-IDENTICAL to the whole-image path and therefore the same picks —
    with halo 512 (what the evaluator uses) and with halo 384, the smallest multiple of 64 above the receptive field
    derived in _tiled_networks (346 px): a halo short of the receptive field could not pass this.  (Window offsets
    must be multiples of 64 for bit-identity: even at scale 32, so that the 2x2 Winograd output tiles of the deepest
    level coincide with the whole image's; other multiples of 32 agree to rounding, next test.)"""
    from spr_pick_amd import DetectionDataset, nms_device, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    den = _filled_denoiser(oracle_state)
    S = 3072
    img = torch.from_numpy(synthetic.micrograph(9, size=S)[0].astype(np.float32) / 255.0).cuda()[None, None]
    eps = torch.randn(img.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    batch = DetectionDataset.make_batch(img, torch.zeros(1, 1))
    with torch.no_grad():
        whole = den.run_pipeline(batch, train=False, eps=eps)
        keep = {k: whole[getattr(P, k)].clone() for k in EVAL_KEYS}
        del whole
        torch.cuda.empty_cache()
        s1, c1 = nms_device(keep["DETECT"][0, 0], 18, 0.02)
        assert len(s1) > 1000
        for tile, halo in ((1024, 512), (1280, 384)):
            torch.cuda.reset_peak_memory_stats()
            tiled = den.run_pipeline(batch, train=False, eps=eps, tile=tile, halo=halo)
            peak = torch.cuda.max_memory_allocated() / 1e9
            for k in EVAL_KEYS:
                a, b = keep[k], tiled[getattr(P, k)]
                assert torch.equal(a, b), "tile %d halo %d: %s differs by %.3e" % (tile, halo, k, float((a - b).abs().max()))
            s2, c2 = nms_device(tiled[P.DETECT][0, 0], 18, 0.02)
            assert torch.equal(c1, c2) and torch.equal(s1, s2)
            print("tile %d halo %d: bit-identical outputs, %d identical picks, peak %.1f GB" % (tile, halo, len(s2), peak))
            del tiled
        with pytest.raises(ValueError):
            den.run_pipeline(batch, train=False, eps=eps, tile=1024, halo=320)     # below the receptive field
    den.unfill()


def test_halo_tiled_inference_on_sizes_off_the_kernel_grid(oracle_state):
    """Windows whose deep U-Net planes are not Winograd-tileable (1920 = 1024 + 2 * 448: 60^2 at scale 32) take the
    direct kernel there: results equal the whole-image path to fp32 rounding (2e-5 of max |value|), and every pick that
    differs is traced to a near-tie or a threshold crossing (tests/pickdiff.py) — nothing else."""
    import pickdiff
    from spr_pick_amd import DetectionDataset, nms_device, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    den = _filled_denoiser(oracle_state)
    S = 2048
    img = torch.from_numpy(synthetic.micrograph(9, size=S)[0].astype(np.float32) / 255.0).cuda()[None, None]
    eps = torch.randn(img.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    batch = DetectionDataset.make_batch(img, torch.zeros(1, 1))
    with torch.no_grad():
        whole = den.run_pipeline(batch, train=False, eps=eps)
        keep = {k: whole[getattr(P, k)].clone() for k in EVAL_KEYS}
        del whole
        torch.cuda.empty_cache()
        tiled = den.run_pipeline(batch, train=False, eps=eps, tile=1024, halo=448)
    den.unfill()
    errs = {k: float((keep[k] - tiled[getattr(P, k)]).abs().max()) / (float(keep[k].abs().max()) + 1e-30) for k in EVAL_KEYS}
    print({k: "%.1e" % v for k, v in errs.items()})
    # the posterior mean and the NLL divide by the predicted variance (near zero on this randomly initialised network),
    # which amplifies a 1e-6 difference of A: 2e-4
    for k, v in errs.items():
        assert v <= (2e-4 if k in ("IMG_DENOISED", "LOSS") else 2e-5), (k, v)
    ma, mb = keep["DETECT"][0, 0], tiled[P.DETECT][0, 0]
    s1, c1 = nms_device(ma, 18, 0.02)
    s2, c2 = nms_device(mb, 18, 0.02)
    e = pickdiff.explain(ma.cpu().numpy(), mb.cpu().numpy(), c1.cpu().numpy(), c2.cpu().numpy(), 18, 0.02)
    print("%d / %d picks; %d + %d differ in %d groups, each with a cause: %s; agreement %.4f" % (
        len(s1), len(s2), len(e["a_only"]), len(e["b_only"]), e["components"], e["roots"][:6], e["jaccard"]))
    assert e["jaccard"] >= 0.99


def test_filled_inference_1024_whole_and_tiled_against_the_oracle(oracle_state):
    """Oracle-level evidence for the filled path above the 128^2 golden case: a 1024^2 synthetic micrograph through
    (a) the whole-image HIP path and (b) the halo-tiled HIP path (tile 256, halo 352: sixteen 960^2 windows) against
    oracle.pipeline.joint_pipeline on the same eps (~15 s of CPU): every output within 1e-4 of its max |value|; the
    HIP NMS on each HIP map equals the C oracle NMS on that map bit for bit; and the picks on the HIP maps differ from
    the picks on the oracle's map only where tests/pickdiff.py finds a near-tie or threshold cause."""
    import pickdiff
    from oracle import nms as onms
    from oracle import pipeline as opipe
    from spr_pick_amd import DetectionDataset, nms_device, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    den = _filled_denoiser(oracle_state)
    S = 1024
    img = torch.from_numpy(synthetic.micrograph(11, size=S)[0].astype(np.float32) / 255.0)[None, None]
    eps = torch.randn(img.shape, generator=torch.Generator().manual_seed(5))
    batch = DetectionDataset.make_batch(img.cuda(), torch.zeros(1, 1))
    with torch.no_grad():
        ref = opipe.joint_pipeline({k: v.clone() for k, v in oracle_state.items()}, img, None, 0, 0, False, eps)
        whole = den.run_pipeline(batch, train=False, eps=eps.cuda())
        tiled = den.run_pipeline(batch, train=False, eps=eps.cuda(), tile=256, halo=352)
    den.unfill()
    want_map = ref["DETECT"][0, 0].numpy()
    s_ref, c_ref = onms.nms_c(want_map, 18, 0.02)
    assert len(s_ref) > 100
    for name, got in (("whole", whole), ("tiled", tiled)):
        for k in EVAL_KEYS:
            close(got[getattr(P, k)], ref[k].numpy(), name="%s %s" % (name, k))
        m = got[P.DETECT][0, 0]
        s, c = nms_device(m, 18, 0.02)
        s_c, c_c = onms.nms_c(m.cpu().numpy(), 18, 0.02)
        assert np.array_equal(c.cpu().numpy(), c_c) and np.array_equal(s.cpu().numpy(), s_c), name
        e = pickdiff.explain(want_map, m.cpu().numpy(), c_ref, c_c, 18, 0.02)
        print("%s: %d picks (oracle map: %d); %d + %d differ, causes %s; agreement %.4f; max |score diff| %.2e" % (
            name, len(s), len(s_ref), len(e["a_only"]), len(e["b_only"]), e["roots"][:4], e["jaccard"], e["delta"]))
        assert e["jaccard"] >= 0.98
