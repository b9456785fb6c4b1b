import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def check_probe(npz, key, tensor, rtol, atol):
    """Compare a tensor with the probe (shape/mean/std/sampled values) stored for it."""
    a = tensor.detach().cpu().numpy().astype(np.float64)
    assert tuple(npz[key + "/shape"]) == a.shape, (key, a.shape)
    flat = a.ravel()
    np.testing.assert_allclose(flat[npz[key + "/idx"]], npz[key + "/val"], rtol=rtol, atol=atol, err_msg=key)
    scale = float(npz[key + "/absmax"]) + 1e-30
    assert abs(flat.mean() - float(npz[key + "/mean"])) <= atol + rtol * scale, key
    assert abs(flat.std() - float(npz[key + "/std"])) <= atol + rtol * scale, key


@pytest.fixture(scope="session")
def oracle_state():
    from oracle import weights
    return weights.make_state(weights.denoiser_shapes(), seed=0)
