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
    assert L.sprk_version() >= 100


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
