"""CPU: the C-ABI library loads and exports every symbol include/sprk.h declares; the product
package refuses to run without the GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sprk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sprk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from spr_pick_amd import _lib
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libsprk.so does not export " + n
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)
    assert L.sprk_version() == _lib.ABI_VERSION


def test_bad_arguments_are_reported_not_crashed():
    from spr_pick_amd import _lib
    L = _lib.lib()
    assert L.sprk_conv2d_fwd(None, None, None, None, None, None, None, 0, None) == -1
    assert b"null" in L.sprk_last_error()
    assert L.sprk_nms2d_ws_bytes(0, 10, 5) == 0
    assert L.sprk_nms2d_ws_bytes(64, 64, 100) > 64 * 64


def test_product_path_has_no_cpu_fallback():
    import spr_pick_amd
    from spr_pick_amd import _lib, cfg, ops, params
    with pytest.raises(_lib.SprkError):
        ops.shift_maxpool2(torch.zeros(1, 1, 4, 4))
    c = cfg.base()
    c[params.ConfigValue.ALGORITHM] = params.NoiseAlgorithm.SELFSUPERVISED_DENOISING
    c[params.ConfigValue.NOISE_STYLE] = "gaussian"
    c[params.ConfigValue.NOISE_VALUE] = params.NoiseValue.UNKNOWN_VARIABLE
    cfg.infer(c, model_only=True)
    with pytest.raises(RuntimeError):
        spr_pick_amd.Denoiser(c, device="cpu", mode="joint")


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "spr_pick_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src or f == "denoiser.py", f


def test_torch_library_registration_and_fake_shapes():
    """Every C entry point of the hot path is a PyTorch custom operator (torch.ops.sprk.*) with a fake implementation:
    shape propagation works without a GPU and without touching libsprk.so."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from spr_pick_amd import ops, torch_ops
    names = set(torch_ops.registered())
    assert {"conv2d_fwd", "conv2d_bwd_data", "conv2d_bwd_weight", "act_bwd", "shift_maxpool2_fwd", "rot4_stack_fwd",
            "unrot4_shift_concat_fwd", "bn_train_fwd", "bn_train_bwd", "bn_eval_fwd", "reparam_fwd", "sigmoid_clamp_fwd",
            "ssdn_fwd", "ssdn_bwd", "nms2d", "pu_loss", "reduce_pending"} <= names
    for n in names:
        assert hasattr(torch.ops.sprk, n)
    with FakeTensorMode():
        x = torch.empty(4, 48, 64, 64, device="cuda")
        w = torch.empty(96, 48, 3, 3, device="cuda")
        g = ops.make_geom(x, None, w, False, 1, 1, (2, 0, 1, 1))
        y = torch.ops.sprk.conv2d_fwd(x, None, w, None, None, None, None, torch_ops.geom_list(g), 0, 1, 1, None)
        assert tuple(y.shape) == (4, 96, 128, 128)          # fused 2x upsampling store
        assert tuple(torch.ops.sprk.conv2d_bwd_data(torch.empty(4, 96, 64, 64, device="cuda"), w,
                                                    torch_ops.geom_list(g), None, 0, None).shape) == (4, 48, 64, 64)
        assert tuple(torch.ops.sprk.shift_maxpool2_fwd(x, 1).shape) == (4, 48, 32, 32)
        assert tuple(torch.ops.sprk.rot4_stack_fwd(torch.empty(2, 1, 8, 8, device="cuda")).shape) == (8, 1, 8, 8)
        assert tuple(torch.ops.sprk.unrot4_shift_concat_fwd(torch.empty(8, 96, 8, 8, device="cuda")).shape) == (2, 384, 8, 8)
        c = torch.empty(32, device="cuda")
        yb, mean, invstd = torch.ops.sprk.bn_train_fwd(torch.empty(8, 32, 9, 9, device="cuda"), c, c, c, c, 0.1, 1e-5, True, 2)
        assert tuple(yb.shape) == (8, 32, 9, 9) and tuple(mean.shape) == (2, 32) == tuple(invstd.shape)   # per-group statistics
        loss, gp = torch.ops.sprk.pu_loss(torch.empty(16, device="cuda"), torch.empty(16, device="cuda"),
                                          torch.empty(17, 17, device="cuda"), 4.0)
        assert tuple(loss.shape) == (1,) and tuple(gp.shape) == (16,)
    # CPU tensors: no CPU kernel is registered, the functional API refuses them up front
    import pytest
    from spr_pick_amd import _lib
    with pytest.raises(_lib.SprkError):
        ops.shift_maxpool2(torch.zeros(1, 1, 4, 4))
