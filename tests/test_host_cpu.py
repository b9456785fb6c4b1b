"""CPU: host-side logic of the product package that needs no GPU (config, module layout,
synthetic data, bench's JSON contract pieces)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN


def test_config_inference_and_name():
    from spr_pick_amd import cfg, params
    c = cfg.base()
    c[params.ConfigValue.ALGORITHM] = params.NoiseAlgorithm.SELFSUPERVISED_DENOISING
    c[params.ConfigValue.NOISE_STYLE] = "gaussian"
    c[params.ConfigValue.NOISE_VALUE] = params.NoiseValue.UNKNOWN_VARIABLE
    cfg.infer(c, model_only=True)
    assert c[params.ConfigValue.PIPELINE] == params.Pipeline.SSDN and c[params.ConfigValue.BLINDSPOT] is True
    assert cfg.config_name(c) == "ssdn-gaussian"
    assert c[params.ConfigValue.TRAIN_PATCH_SIZE] == 64 and c[params.ConfigValue.NMS] == 15


def test_module_parameter_names_match_reference_layout():
    from spr_pick_amd import networks
    layout = json.load(open(os.path.join(GOLDEN, "state_layout.json")))
    jn = networks.JointNetwork(1, 2, blindspot=True, detect=True)
    sg = networks.DualNetworkShallow(1, 1)
    ours = {"models.denoiser_model." + k: list(v.shape) for k, v in jn.state_dict().items()}
    ours.update({"models.sigma_estimation_model." + k: list(v.shape) for k, v in sg.state_dict().items()})
    ref = {k: v for k, v in layout.items() if k.startswith("models.")}
    assert ours == ref
    assert jn.detector.detector.width == 63 and jn.input_wh_mul() == 32
    assert jn.fill() == 4
    assert [m.dilation for m in jn.detector.detector.features.features] == [1, 2, 2, 4, 4]
    jn.unfill()
    assert [m.dilation for m in jn.detector.detector.features.features] == [1, 1, 1, 1, 1]


def test_reference_init_statistics():
    """kaiming_normal_(a=0.1) / zero bias for the U-Nets (joint_network_v2.py:176-187)."""
    from spr_pick_amd import networks
    torch.manual_seed(0)
    net = networks.DualNetwork(1, 2, blindspot=True, detect=True)
    w = net.decode_block_1[2].weight
    assert abs(w.std().item() - np.sqrt(2 / 1.01 / (96 * 9))) < 2e-3
    assert float(net.decode_block_1[2].bias.abs().max()) == 0.0


def test_synthetic_micrographs_are_deterministic():
    from spr_pick_amd import synthetic
    a, ca, la = synthetic.micrograph(3, size=256)
    b, cb, lb = synthetic.micrograph(3, size=256)
    assert a.dtype == np.uint8 and a.shape == (256, 256) and np.array_equal(a, b) and np.array_equal(la, lb)
    batches = synthetic.patch_batches(2, 4, [synthetic.micrograph(0, size=512)], device="cpu")
    inp, tgt = batches[0]
    assert inp.shape == (4, 1, 64, 64) and tgt.shape == (4, 1)
    assert float(inp.min()) >= 0 and float(inp.max()) <= 1
    assert set(np.unique(np.sign(tgt.numpy()))) <= {-1.0, 1.0}


def test_pu_loss_matches_oracle():
    from oracle import pipeline
    from spr_pick_amd.denoiser import PuLoss
    g = torch.Generator().manual_seed(0)
    p = torch.rand(16, 1, 1, 1, generator=g) * 0.98 + 0.01
    y = torch.tensor([1.0, -1, -1, 0.3, -1, -1, -1, 0.0, -1, -1, -1, -1, 1.0, -1, -1, -1]).reshape(16, 1)
    want = pipeline.pu_loss(0.01, p, y)
    got = PuLoss().mask_form(0.01, p, y)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
    y2 = torch.full((16, 1), -1.0)
    assert torch.allclose(PuLoss().mask_form(0.05, p, y2), pipeline.pu_loss(0.05, p, y2), rtol=1e-5, atol=1e-6)


def test_pick_filter_and_writer(tmp_path):
    """N2 (train.py:563-571): border filter is strict (> 30, < size-30) and columns are (row, col)."""
    from spr_pick_amd import picks
    scores = np.asarray([0.9, 0.8, 0.7, 0.6, 0.5], dtype=np.float32)
    coords = np.asarray([[50, 40], [30, 40], [31, 31], [100, 69], [69, 100]], dtype=np.int32)  # (col, row)
    s, c = picks.filter_picks(scores, coords, (100, 140))
    # rows must be in (30, 70), cols in (30, 110)
    assert c.tolist() == [[50, 40], [31, 31], [100, 69]]
    p = tmp_path / "m_scores.txt"
    assert picks.write_scores(str(p), "mic", scores, coords, (100, 140)) == 3
    lines = p.read_text().splitlines()
    assert lines[0] == "image_name\tx_coord\ty_coord\tscore"
    assert lines[1] == "mic\t40\t50\t" + str(np.float32(0.9))
