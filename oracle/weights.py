"""Deterministic, storage-free weights for parity tests (TEST INFRASTRUCTURE ONLY).

The reference holds no golden weights, and a seeded ``state_dict`` of the joint
model is 9 MB — too big for a fixture.  Instead every tensor is generated from a
NumPy PCG64 stream keyed by (seed, crc32(parameter name)), so the fixture
generator (which loads them into the *reference* modules), the CPU restatement
and the HIP path all see bit-identical parameters without storing them.

Names and shapes follow the reference's ``state_dict`` layout
(reference ctor code: spr_pick/models/joint_network_v2.py:12-159,
:543-547; joint_network_v2_shallow.py:12-168; feature_extractor.py:102-144,
:279-290, :326-346; classifier.py:8-11).
Biases / BN statistics are drawn non-trivially (the reference initialises them
to 0 / 1) so that the bias, BN-affine and running-stat paths are exercised.
"""
import zlib

import numpy as np
import torch


def unet_shapes(prefix="", in_ch=1, out_ch=2, detect=True):
    """DualNetwork(in, out, blindspot=True, detect=True) parameter shapes."""
    s = {}

    def conv(name, co, ci, k):
        s[prefix + name + ".weight"] = (co, ci, k, k)
        s[prefix + name + ".bias"] = (co,)

    conv("encode_block_1.0", 48, in_ch, 3)
    conv("encode_block_1.2", 48, 48, 3)
    for i in (2, 3, 4, 5, 6):
        conv("encode_block_%d.0" % i, 48, 48, 3)
    conv("decode_block_5.0", 96, 96, 3)
    conv("decode_block_5.2", 96, 96, 3)
    for i in (4, 3, 2):
        conv("decode_block_%d.0" % i, 96, 144, 3)
        conv("decode_block_%d.2" % i, 96, 96, 3)
    conv("decode_block_1.0", 96, 96 + in_ch, 3)
    conv("decode_block_1.2", 96, 96, 3)
    conv("output_block.0", 384, 384, 1)
    conv("output_block.2", 96, 384, 1)
    if detect:
        conv("output_conv_f", 1, 96, 1)
    conv("output_conv", out_ch, 96, 1)
    return s


def sigma_shapes(prefix="", in_ch=1, out_ch=1):
    """DualNetworkShallow(in, out, blindspot=False, detect=False) parameter shapes."""
    s = {}

    def conv(name, co, ci, k):
        s[prefix + name + ".weight"] = (co, ci, k, k)
        s[prefix + name + ".bias"] = (co,)

    conv("encode_block_1.0", 48, in_ch, 3)
    conv("encode_block_1.2", 48, 48, 3)
    for i in (2, 3, 6):
        conv("encode_block_%d.0" % i, 48, 48, 3)
    conv("decode_block_5.0", 96, 96, 3)
    conv("decode_block_5.2", 96, 96, 3)
    for i in (3, 2):
        conv("decode_block_%d.0" % i, 96, 144, 3)
        conv("decode_block_%d.2" % i, 96, 96, 3)
    conv("decode_block_1.0", 96, 96 + in_ch, 3)
    conv("decode_block_1.2", 96, 96, 3)
    conv("output_block.0", 96, 96, 1)
    conv("output_block.2", 96, 96, 1)
    conv("detect_block.0", 96, 96, 1)
    conv("detect_block.2", 96, 96, 1)
    conv("output_conv", out_ch, 96, 1)
    conv("output_conv_f", 1, 96, 1)
    return s


def _bn(s, name, c):
    s[name + ".weight"] = (c,)
    s[name + ".bias"] = (c,)
    s[name + ".running_mean"] = (c,)
    s[name + ".running_var"] = (c,)
    s[name + ".num_batches_tracked"] = ()


def detector_shapes(prefix=""):
    """Detector = BatchNorm2d(1) + LinearClassifier(ResNet8(bn=True))."""
    s = {}
    _bn(s, prefix + "m", 1)
    f = prefix + "detector.features.features."
    s[f + "0.conv.weight"] = (32, 1, 7, 7)
    _bn(s, f + "0.bn", 32)
    for idx, (nin, nh, nout) in ((1, (32, 32, 32)), (2, (32, 32, 64)), (3, (64, 64, 64))):
        p = f + "%d." % idx
        if nin != nout:
            s[p + "proj.weight"] = (nout, nin, 1, 1)
        s[p + "conv0.weight"] = (nh, nin, 3, 3)
        _bn(s, p + "bn0", nh)
        s[p + "conv1.weight"] = (nout, nh, 3, 3)
        _bn(s, p + "bn1", nout)
    s[f + "4.conv.weight"] = (128, 64, 3, 3)
    _bn(s, f + "4.bn", 128)
    s[prefix + "detector.classifier.weight"] = (1, 128, 1, 1)
    s[prefix + "detector.classifier.bias"] = (1,)
    return s


def joint_shapes(prefix=""):
    s = unet_shapes(prefix + "denoise_branch.")
    s.update(detector_shapes(prefix + "detector."))
    return s


def denoiser_shapes():
    """All entries of Denoiser.state_dict(params_only=True) under ``models.``."""
    s = joint_shapes("denoiser_model.")
    s.update(sigma_shapes("sigma_estimation_model."))
    return s


def make_tensor(name, shape, seed=0):
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_var":
        a = rng.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_mean":
        a = rng.normal(0.0, 0.1, size=shape)
    elif leaf == "weight" and len(shape) == 1:  # BN gamma
        a = rng.uniform(0.5, 1.5, size=shape)
    elif leaf == "bias":
        a = rng.normal(0.0, 0.05, size=shape)
    else:  # conv weight: kaiming-normal for leaky slope 0.1
        fan_in = shape[1] * shape[2] * shape[3]
        a = rng.normal(0.0, np.sqrt(2.0 / (1.0 + 0.01) / fan_in), size=shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def make_state(shapes, seed=0):
    return {k: make_tensor(k, v, seed) for k, v in shapes.items()}
