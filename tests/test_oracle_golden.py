"""CPU: the oracle restatement (oracle/) against golden vectors generated from the
reference itself (oracle/gen_golden.py).  This is what pins the oracle."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, check_probe, golden
from oracle import networks, nms, pipeline, weights

MODEL = pipeline.MODEL
# fp32 restatement vs fp32 reference on the same CPU kernels: differences come only from
# op ordering (e.g. F.pad+conv vs pad/conv/crop), so tolerances are tight.
RTOL, ATOL = 1e-5, 1e-6


def clone_state(sd):
    return {k: v.clone() for k, v in sd.items()}


def test_state_layout_matches_reference():
    layout = json.load(open(os.path.join(GOLDEN, "state_layout.json")))
    ours = weights.denoiser_shapes()
    for top in ("models.", "_models."):
        ref = {k[len(top):]: v for k, v in layout.items() if k.startswith(top)}
        assert set(ref) == set(ours)
        for k, shp in ref.items():
            assert tuple(shp) == tuple(ours[k]), k
    assert layout["cfg"] == "cfg"
    assert len(layout) == 2 * len(ours) + 1 == 273


def test_unet_forward(oracle_state):
    g = golden("unet_fwd.npz")
    taps = {}
    with torch.no_grad():
        out = networks.unet_blindspot(oracle_state, MODEL + "denoise_branch.", torch.from_numpy(g["x"]), taps)
    np.testing.assert_allclose(out.numpy(), g["out_stats"], rtol=RTOL, atol=ATOL)
    for ours, theirs in (("pool1", "encode_block_1"), ("pool2", "encode_block_2"), ("pool3", "encode_block_3"),
                         ("pool4", "encode_block_4"), ("pool5", "encode_block_5"), ("encoded", "encode_block_6"),
                         ("decoded", "decode_block_1")):
        check_probe(g, theirs, taps[ours], RTOL, ATOL)


def test_blindspot_property(oracle_state):
    g = golden("unet_fwd.npz")
    assert bool(g["blindspot_same"])
    x = torch.from_numpy(g["x"])
    x2 = x.clone()
    x2[0, 0, 20, 37] += 0.25
    with torch.no_grad():
        a = networks.unet_blindspot(oracle_state, MODEL + "denoise_branch.", x)
        b = networks.unet_blindspot(oracle_state, MODEL + "denoise_branch.", x2)
    assert torch.equal(a[0, :, 20, 37], b[0, :, 20, 37])
    assert not torch.equal(a[0], b[0])


def test_sigma_net_and_detector(oracle_state):
    g = golden("parts.npz")
    sd = clone_state(oracle_state)
    with torch.no_grad():
        y = networks.sigma_unet(sd, pipeline.SIGMA, torch.from_numpy(g["x"]))
        np.testing.assert_allclose(y.numpy(), g["sigma_out"], rtol=RTOL, atol=ATOL)
        z, zf = torch.from_numpy(g["z"]), torch.from_numpy(g["zf"])
        d = networks.detector(sd, MODEL + "detector.", z, filled=False, training=False)
        np.testing.assert_allclose(d.numpy(), g["det_eval_unfilled"], rtol=1e-4, atol=1e-5)
        d = networks.detector(sd, MODEL + "detector.", zf, filled=True, training=False)
        np.testing.assert_allclose(d.numpy(), g["det_eval_filled"], rtol=1e-4, atol=1e-5)
        d = networks.detector(sd, MODEL + "detector.", z[:1], filled=True, training=False)
        np.testing.assert_allclose(d.numpy(), g["det_eval_filled64"], rtol=1e-4, atol=1e-5)
        # filled map at (31,31) is the unfilled score of the same 64x64 patch (Topaz fill trick)
        np.testing.assert_allclose(d[0, 0, 31, 31].item(), g["det_eval_unfilled"][0, 0, 0, 0], rtol=1e-4, atol=1e-5)
        d = networks.detector(sd, MODEL + "detector.", z, filled=False, training=True)
        np.testing.assert_allclose(d.numpy(), g["det_train_unfilled"], rtol=1e-4, atol=1e-5)
    for k in g.files:
        if k.startswith("bn_after/"):
            ours = sd[MODEL + "detector." + k[len("bn_after/"):]]
            np.testing.assert_allclose(ours.numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    assert int(g["fill_stride"]) == 4


@pytest.mark.parametrize("tag", ["w", "h"])
def test_joint_train_step(oracle_state, tag):
    g = golden("joint_train_%s.npz" % tag)
    sd = clone_state(oracle_state)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    res = pipeline.joint_pipeline(sd, torch.from_numpy(g["inp"]), torch.from_numpy(g["target"]),
                                  float(g["alpha"]), float(g["tau"]), True,
                                  torch.from_numpy(g["eps"]), torch.from_numpy(g["eps_flip"]), float(g["flip_p"]))
    res["LOSS"].mean().backward()
    for k in ("LOSS", "DENOISE_LOSS", "DETECT_LOSS", "AUG_LOSS", "DETECT", "IMG_MU", "IMG_DENOISED",
              "NOISE_STD_DEV", "MODEL_STD_DEV"):
        np.testing.assert_allclose(res[k].detach().numpy(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)
    nograd = set(g["nograd"].tolist())
    assert len(nograd) == 12
    for name, p in sd.items():
        if not p.requires_grad:
            continue
        if name in nograd:
            assert p.grad is None, name
        else:
            check_probe(g, "grad/" + name, p.grad, 2e-4, 1e-7)
            np.testing.assert_allclose(p.grad.double().norm().item(), float(g["grad/" + name + "/norm"]), rtol=1e-4)
    for k in g.files:
        if k.startswith("bn_after/"):
            ours = sd[MODEL + "detector." + k[len("bn_after/"):]]
            np.testing.assert_allclose(ours.detach().numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_joint_eval_and_picks(oracle_state):
    g = golden("joint_eval.npz")
    sd = clone_state(oracle_state)
    with torch.no_grad():
        res = pipeline.joint_pipeline(sd, torch.from_numpy(g["inp"]), None, 0, 0, False, torch.from_numpy(g["eps"]))
    for k in ("LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        np.testing.assert_allclose(res[k].numpy(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)
    # NMS on the REFERENCE's score map: coordinates must be identical
    for r in (18, 5):
        s, c = nms.nms_literal(g["DETECT"][0, 0], r, 0.02)
        assert np.array_equal(c, g["nms%d_coords" % r])
        assert np.array_equal(s, g["nms%d_scores" % r])


def test_ssdn_pipeline(oracle_state):
    g = golden("ssdn_eval.npz")
    with torch.no_grad():
        res = pipeline.ssdn_pipeline(clone_state(oracle_state), torch.from_numpy(g["inp"]))
    for k in ("LOSS", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        np.testing.assert_allclose(res[k].numpy(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)


def nms_case_names():
    g = golden("nms_cases.npz")
    return sorted({k.split("/")[0] for k in g.files})


@pytest.mark.parametrize("name", nms_case_names())
@pytest.mark.parametrize("impl", ["literal", "c"])
def test_nms_cases(name, impl):
    g = golden("nms_cases.npz")
    fn = nms.nms_literal if impl == "literal" else nms.nms_c
    s, c = fn(g[name + "/x"], int(g[name + "/r"]), float(g[name + "/thr"]))
    assert c.dtype == np.int32 and s.dtype == np.float32
    assert np.array_equal(c, g[name + "/coords"].reshape(-1, 2)), name
    assert np.array_equal(s, g[name + "/scores"]), name


def test_nms_wrap_quirk_is_exercised():
    """The x-overflow wrap (clip to W, not W-1) is part of the contract: check the fixture
    really distinguishes it from a plain clipped disk."""
    g = golden("nms_cases.npz")
    c = g["wrap_r5/coords"]
    picked = {(int(x), int(y)) for x, y in c}
    assert (38, 10) in picked
    assert (0, 11) not in picked and (0, 14) not in picked   # suppressed only via the wrap
    assert (0, 16) in picked                                  # outside the wrapped footprint
    c = g["wrap_corner_r3/coords"]
    picked = {(int(x), int(y)) for x, y in c}
    assert (23, 0) in picked and (0, 1) not in picked


def test_lr_schedule_and_width():
    g = golden("misc.npz")
    lr = [pipeline.trainer_lrate(int(i), 80000) for i in g["lr_iters"]]
    np.testing.assert_allclose(lr, g["lr"], rtol=1e-12, atol=0)
    assert int(g["det_width"]) == networks.DET_WIDTH


def test_joint_poisson_step_and_eval(oracle_state):
    """The poisson likelihood branch (denoiser_v2.py:412-424) against the reference's own outputs
    (oracle/gen_golden_poisson.py): a joint train step with gradients and a filled eval pass."""
    g = golden("joint_poisson.npz")
    sd = clone_state(oracle_state)
    sd[MODEL + "denoise_branch.output_conv.bias"][0] += float(g["mu_bias"])
    sd0 = clone_state(sd)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    res = pipeline.joint_pipeline(sd, torch.from_numpy(g["inp"]), torch.from_numpy(g["target"]), float(g["alpha"]),
                                  float(g["tau"]), True, torch.from_numpy(g["eps"]), torch.from_numpy(g["eps_flip"]),
                                  float(g["flip_p"]), noise_style="poisson")
    res["LOSS"].mean().backward()
    assert res["NOISE_STD_DEV"].shape == (3, 64, 64)            # per pixel now, not per image
    for k in ("LOSS", "DENOISE_LOSS", "DETECT_LOSS", "AUG_LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV",
              "MODEL_STD_DEV"):
        np.testing.assert_allclose(res[k].detach().numpy(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)
    assert 0.2 < float(g["frac_mu_below_floor"]) < 0.8          # both sides of max(mu, 1e-3) are exercised
    n = 0
    for name, p in sd.items():
        if p.requires_grad and ("grad/" + name + "/val") in g.files:
            # (the likelihood of this fixture is large — loss ~90 — and so are its gradients: absolute slack scales with them)
            check_probe(g, "grad/" + name, p.grad, 2e-4, 1e-6 * float(g["grad/" + name + "/absmax"]) + 1e-7)
            n += 1
    assert n > 60
    with torch.no_grad():
        ev = pipeline.joint_pipeline(sd0, torch.from_numpy(g["eval/inp"]), None, 0, 0, False,
                                     torch.from_numpy(g["eval/eps"]), noise_style="poisson")
    for k in ("LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        np.testing.assert_allclose(ev[k].numpy(), g["eval/" + k], rtol=2e-5, atol=2e-6, err_msg=k)
