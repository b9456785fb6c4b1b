"""GPU parity of every libsprk.so operator against a plain PyTorch fp32/fp64 CPU statement of the
same op (floating-point kernels: tolerance stated per test), through the C ABI."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# fp32 MFMA is a k-ordered fp32 fma chain; the CPU reference below is evaluated in fp64, so the
# budget is a few fp32 ulps of the accumulated magnitude: 2e-5 relative to the tensor's max |value|.
REL = 2e-5


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def close(got, want, rel=REL, name=""):
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs()
    worst = err.max().item()
    if not worst <= rel * scale:
        idx = np.unravel_index(int(err.argmax()), err.shape)
        raise AssertionError("%s: max err %.3e (scale %.3e, rel %.2e) at %s got %.6g want %.6g; mismatches %d / %d" % (
            name, worst, scale, worst / scale, idx, got[idx].item(), want[idx].item(),
            int((err > rel * scale).sum()), err.numel()))


def ref_conv(x, x2, w, b, up1, stride, dil, pad, act):
    """fp64 CPU statement: cat(up(x), x2) -> zero pad (t,b,l,r) -> conv -> +bias -> act."""
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up1 else x
    if x2 is not None:
        xin = torch.cat((xin, x2), 1)
    pt, pb, pl, pr = pad
    xin = F.pad(xin, (pl, pr, pt, pb))
    y = F.conv2d(xin, w, b, stride=stride, dilation=dil)
    if act == 1:
        y = F.leaky_relu(y, 0.1)
    elif act == 2:
        y = F.relu(y)
    return y


CONV_CASES = [
    # name, N, C1, C2, H(in, after upsample), W, up1, Cout, K, stride, dil, pad(t,b,l,r), act, bias
    ("enc1.0 shift 1->48 @64", 8, 1, 0, 64, 64, 0, 48, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("enc1.2 shift 48->48 @64", 4, 48, 0, 64, 64, 0, 48, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("enc2 shift 48->48 @32", 8, 48, 0, 32, 32, 0, 48, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("enc4 shift 48->48 @8", 8, 48, 0, 8, 8, 0, 48, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("enc6 shift 48->48 @2", 8, 48, 0, 2, 2, 0, 48, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("dec5.0 shift up(48)+48->96 @4", 8, 48, 48, 4, 4, 1, 96, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("dec4.0 shift up(96)+48->96 @8", 8, 96, 48, 8, 8, 1, 96, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("dec3.2 shift 96->96 @16", 8, 96, 0, 16, 16, 0, 96, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("dec1.0 shift up(96)+1->96 @64", 4, 96, 1, 64, 64, 1, 96, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("dec1.2 shift 96->96 @64", 4, 96, 0, 64, 64, 0, 96, 3, 1, 1, (2, 0, 1, 1), 1, True),
    ("sigma plain 48->48 @32", 4, 48, 0, 32, 32, 0, 48, 3, 1, 1, (1, 1, 1, 1), 1, True),
    ("sigma plain up(96)+1->96 @64", 2, 96, 1, 64, 64, 1, 96, 3, 1, 1, (1, 1, 1, 1), 1, True),
    ("head 1x1 384->384 @64", 2, 384, 0, 64, 64, 0, 384, 1, 1, 1, (0, 0, 0, 0), 1, True),
    ("head 1x1 384->96 @64", 2, 384, 0, 64, 64, 0, 96, 1, 1, 1, (0, 0, 0, 0), 1, True),
    ("head 1x1 96->2 @64", 2, 96, 0, 64, 64, 0, 2, 1, 1, 1, (0, 0, 0, 0), 0, True),
    ("sigma 1x1 96->1 @64", 2, 96, 0, 64, 64, 0, 1, 1, 1, 1, (0, 0, 0, 0), 0, True),
    ("det conv7 s2 1->32 @64", 4, 1, 0, 64, 64, 0, 32, 7, 2, 1, (0, 0, 0, 0), 0, False),
    ("det 3x3 valid 32->32 @29", 4, 32, 0, 29, 29, 0, 32, 3, 1, 1, (0, 0, 0, 0), 0, False),
    ("det 3x3 d2 32->32 @27", 4, 32, 0, 27, 27, 0, 32, 3, 1, 2, (0, 0, 0, 0), 0, False),
    ("det 3x3 d2 s2 32->64 @21", 4, 32, 0, 21, 21, 0, 64, 3, 2, 2, (0, 0, 0, 0), 0, False),
    ("det proj 1x1 s2 32->64 @17", 4, 32, 0, 17, 17, 0, 64, 1, 2, 1, (0, 0, 0, 0), 0, False),
    ("det 3x3 64->128 @3", 4, 64, 0, 3, 3, 0, 128, 3, 1, 1, (0, 0, 0, 0), 2, False),
    ("det cls 1x1 128->1 @1", 4, 128, 0, 1, 1, 0, 1, 1, 1, 1, (0, 0, 0, 0), 0, True),
    ("filled conv7 pad31 1->32 @40x56", 1, 1, 0, 40, 56, 0, 32, 7, 1, 1, (31, 31, 31, 31), 0, False),
    ("filled 3x3 d4 32->64 @60x44", 1, 32, 0, 60, 44, 0, 64, 3, 1, 4, (0, 0, 0, 0), 0, False),
    ("filled 3x3 d8 64->64 @48x52", 1, 64, 0, 48, 52, 0, 64, 3, 1, 8, (0, 0, 0, 0), 0, False),
    ("odd sizes 5->7 3x3 @13x9 pad(1,2,0,1)", 3, 5, 0, 13, 9, 0, 7, 3, 1, 1, (1, 2, 0, 1), 1, True),
]


@pytest.mark.parametrize("naive", [0, 1], ids=["mfma", "direct"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_fwd_bwd(case, naive):
    from spr_pick_amd import _lib, ops
    name, N, C1, C2, H, W, up1, Cout, K, stride, dil, pad, act, has_b = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    h1, w1 = (H // 2, W // 2) if up1 else (H, W)
    x = torch.randn(N, C1, h1, w1, generator=g)
    x2 = torch.randn(N, C2, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, C1 + C2, K, K, generator=g) / np.sqrt((C1 + C2) * K * K)
    b = torch.randn(Cout, generator=g) * 0.1 if has_b else None
    d = dev()
    L = _lib.lib()
    try:
        dl = [t.to(d).requires_grad_(True) if t is not None else None for t in (x, x2, w, b)]
        # "direct": SPRK_DT_NAIVE in the geometry's dtype routes this call (forward and both backward kernels) to the
        # plain per-output-element kernels — a per-call flag, the library keeps no mode
        y = ops.conv2d(dl[0], dl[2], dl[3], x2=dl[1], up1=bool(up1), stride=stride, dil=dil, pad=pad, act=act,
                       dtype=_lib.DT_NAIVE if naive else 0)
        # fp64 CPU reference.  The backward of (Leaky)ReLU depends on the SIGN of the pre-activation;
        # an fp32 result within rounding of zero may legitimately land on the other side than the
        # fp64 one, so the reference backward uses the sign pattern of the GPU output.
        leaves = [t.double().requires_grad_(True) if t is not None else None for t in (x, x2, w, b)]
        pre = ref_conv(leaves[0], leaves[1], leaves[2], leaves[3], up1, stride, dil, pad, 0)
        yr_true = ref_conv(leaves[0], leaves[1], leaves[2], leaves[3], up1, stride, dil, pad, act)
        close(y, yr_true, name=name + " y")
        if act:
            pos = (y.detach().cpu() > 0)
            yr = torch.where(pos, pre, pre * (0.1 if act == 1 else 0.0))
        else:
            yr = pre
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy.double())
        y.backward(gy.to(d))
        torch.cuda.synchronize()
    finally:
        pass
    # gradients sum over up to N*H*W terms: budget relative to the gradient's own scale
    close(dl[0].grad, leaves[0].grad, rel=5e-5, name=name + " gx")
    if x2 is not None:
        close(dl[1].grad, leaves[1].grad, rel=5e-5, name=name + " gx2")
    close(dl[2].grad, leaves[2].grad, rel=5e-5, name=name + " gw")
    if b is not None:
        close(dl[3].grad, leaves[3].grad, rel=5e-5, name=name + " gb")


WINO_CASES = [
    # name, N, C1, C2, H, W, Cout, pad(t,b,l,r), act, bias, backward-data on the Winograd kernel, backward-weight too
    # -- every forward is large enough (>= 192 workgroups) for the Winograd F(2x2,3x3) kernel; the
    # backward-data correlation of 96+1 inputs (97 output channels = 2 half-empty groups) stays direct; the
    # Winograd backward-weight kernel takes 81-96 output channels, sources of 48 k (+ <= 16) channels and
    # >= 8192 regions of 4x16 pixels
    ("wino shift 96->96 @64", 12, 96, 0, 64, 64, 96, (2, 0, 1, 1), 1, True, True, False),
    ("wino shift 96+48->96 @32", 48, 96, 48, 32, 32, 96, (2, 0, 1, 1), 1, True, True, False),
    ("wino plain 96+1->96 @64", 12, 96, 1, 64, 64, 96, (1, 1, 1, 1), 1, True, False, False),
    ("wino shift 48->48 @64", 12, 48, 0, 64, 64, 48, (2, 0, 1, 1), 1, True, True, False),
    ("wino ragged 90->70 @8x96", 64, 90, 0, 8, 96, 70, (1, 1, 1, 1), 0, False, True, False),
    ("wino up-shift 41->96 @16x32 pad(0,2,2,0)", 96, 41, 0, 16, 32, 96, (0, 2, 2, 0), 2, True, True, False),
    ("wino square tiles 96+48->96 @16x16", 192, 96, 48, 16, 16, 96, (2, 0, 1, 1), 1, True, True, False),
    ("wino square tiles 48->48 @32x48", 32, 48, 0, 32, 48, 48, (1, 1, 1, 1), 2, True, True, False),
    ("wino+wgrad shift 96+1->96 @64", 128, 96, 1, 64, 64, 96, (2, 0, 1, 1), 1, True, False, True),
    ("wino+wgrad shift 96+48->96 @32x128", 128, 96, 48, 32, 128, 96, (2, 0, 1, 1), 1, True, True, True),
    ("wino+wgrad plain 48+10->88 @16x256 pad(1,1,2,0)", 130, 48, 10, 16, 256, 88, (1, 1, 2, 0), 0, True, False, True),
    # the backward-weight kernel's border masks: two region columns (first / last only), one column (first and last),
    # two region rows
    ("wino+wgrad shift 96->96 @32x32 (two region columns)", 128, 96, 0, 32, 32, 96, (2, 0, 1, 1), 1, True, True, True),
    ("wino+wgrad plain 48->96 @64x16 (one region column)", 128, 48, 0, 64, 16, 96, (1, 1, 1, 1), 0, True, True, True),
    ("wino+wgrad shift 96->88 @8x96 (two region rows)", 172, 96, 0, 8, 96, 88, (2, 0, 1, 1), 2, False, True, True),
]


@pytest.mark.parametrize("case", WINO_CASES, ids=[c[0] for c in WINO_CASES])
def test_conv2d_winograd(case):
    """The 3x3 stride-1 layers wide enough for the Winograd kernel: forward and all three gradients against
    the fp64 CPU convolution, with the same budgets as the direct MFMA kernel (the transform adds a few fp32
    roundings per product, far inside 2e-5 of the tensor scale), plus proof that the Winograd kernel ran."""
    from spr_pick_amd import _lib, ops
    name, N, C1, C2, H, W, Cout, pad, act, has_b, bwd_wino, wg_wino = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    x = torch.randn(N, C1, H, W, generator=g)
    x2 = torch.randn(N, C2, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / np.sqrt((C1 + C2) * 9)
    b = torch.randn(Cout, generator=g) * 0.1 if has_b else None
    d = dev()
    L = _lib.lib()
    dl = [t.to(d).requires_grad_(True) if t is not None else None for t in (x, x2, w, b)]
    before = L.sprk_wino_launch_count()
    y = ops.conv2d(dl[0], dl[2], dl[3], x2=dl[1], pad=pad, act=act)
    assert L.sprk_wino_launch_count() == before + 1, "forward did not take the Winograd kernel"
    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (x, x2, w, b)]
    pre = ref_conv(leaves[0], leaves[1], leaves[2], leaves[3], 0, 1, 1, pad, 0)
    close(y, ref_conv(leaves[0], leaves[1], leaves[2], leaves[3], 0, 1, 1, pad, act), name=name + " y")
    yr = torch.where(y.detach().cpu() > 0, pre, pre * (0.1 if act == 1 else 0.0)) if act else pre
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    y.backward(gy.to(d))
    torch.cuda.synchronize()
    assert L.sprk_wino_launch_count() == before + 1 + int(bwd_wino) + int(wg_wino), "backward kernel choice"
    close(dl[0].grad, leaves[0].grad, rel=5e-5, name=name + " gx")
    if x2 is not None:
        close(dl[1].grad, leaves[1].grad, rel=5e-5, name=name + " gx2")
    close(dl[2].grad, leaves[2].grad, rel=5e-5, name=name + " gw")
    if b is not None:
        close(dl[3].grad, leaves[3].grad, rel=5e-5, name=name + " gb")


@pytest.mark.parametrize("shape", [(64, 96, 0, 32, 32, 96), (192, 96, 48, 16, 16, 96), (12, 48, 0, 64, 64, 48)])
def test_conv2d_winograd_fused_upsample(shape):
    """up_out on the Winograd kernel (the decoder's second convolutions write their nearest-x2 upsampled output
    directly): every value appears in its 2x2 block, values and gradients as conv + nn.Upsample in fp64."""
    from spr_pick_amd import _lib, ops
    N, C1, C2, H, W, Cout = shape
    g = torch.Generator().manual_seed(31 + H)
    x = torch.randn(N, C1, H, W, generator=g)
    x2 = torch.randn(N, C2, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / np.sqrt((C1 + C2) * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    pad = (2, 0, 1, 1)
    d = dev()
    L = _lib.lib()
    dl = [t.to(d).requires_grad_(True) if t is not None else None for t in (x, x2, w, b)]
    before = L.sprk_wino_launch_count()
    y = ops.conv2d(dl[0], dl[2], dl[3], x2=dl[1], pad=pad, act=1, up_out=True)
    assert L.sprk_wino_launch_count() == before + 1, "up_out forward did not take the Winograd kernel"
    assert tuple(y.shape) == (N, Cout, 2 * H, 2 * W)
    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (x, x2, w, b)]
    yr = F.interpolate(ref_conv(leaves[0], leaves[1], leaves[2], leaves[3], 0, 1, 1, pad, 1), scale_factor=2, mode="nearest")
    close(y, yr, name="wino up_out y")
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    y.backward(gy.to(d))
    close(dl[0].grad, leaves[0].grad, rel=5e-5, name="wino up_out gx")
    close(dl[2].grad, leaves[2].grad, rel=5e-5, name="wino up_out gw")
    close(dl[3].grad, leaves[3].grad, rel=5e-5, name="wino up_out gb")


def test_conv2d_winograd_affine_epilogue_and_switch():
    """Inference epilogue relu(conv * scale + shift) on the Winograd kernel, and SPRK_WINO-independent
    agreement with the plain direct kernel on the same input (SPRK_DT_NAIVE routes around both)."""
    from spr_pick_amd import _lib, ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 96, 96, 128, generator=g)
    w = torch.randn(96, 96, 3, 3, generator=g) / 29
    scale, shift = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g)
    want = F.relu(F.conv2d(F.pad(x.double(), (1, 1, 2, 0)), w.double()) * scale.double().view(1, -1, 1, 1)
                  + shift.double().view(1, -1, 1, 1))
    d = dev()
    L = _lib.lib()
    xd, wd = x.to(d), w.to(d)
    geom = ops.make_geom(xd, None, wd, False, 1, 1, (2, 0, 1, 1))
    before = L.sprk_wino_launch_count()
    y = ops.conv2d_forward(xd, None, wd, geom, act=ops.ACT_RELU, scale=scale.to(d), shift=shift.to(d))
    assert L.sprk_wino_launch_count() == before + 1
    close(y, want, name="wino affine epilogue")
    geom_direct = ops.make_geom(xd, None, wd, False, 1, 1, (2, 0, 1, 1), dtype=_lib.DT_NAIVE)
    y2 = ops.conv2d_forward(xd, None, wd, geom_direct, act=ops.ACT_RELU, scale=scale.to(d), shift=shift.to(d))
    assert L.sprk_wino_launch_count() == before + 1
    close(y, y2, name="wino vs direct kernel")


def test_conv_fused_upsample_output():
    """conv + LeakyReLU + nn.Upsample(2, nearest) fused into the stores, and its backward."""
    from spr_pick_amd import ops
    g = torch.Generator().manual_seed(9)
    for shape in ((4, 48, 8, 8), (3, 96, 32, 32), (2, 5, 2, 2)):
        x = torch.randn(shape, generator=g)
        w = torch.randn(24, shape[1], 3, 3, generator=g) / 20
        b = torch.randn(24, generator=g) * 0.1
        d = dev()
        xd, wd, bd = (t.to(d).requires_grad_(True) for t in (x, w, b))
        y = ops.conv2d(xd, wd, bd, pad=(2, 0, 1, 1), act=1, up_out=True)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        pre = F.conv2d(F.pad(xr, (1, 1, 2, 0)), wr, br)
        close(y, F.interpolate(F.leaky_relu(pre, 0.1), scale_factor=2, mode="nearest"), name="up_out y")
        pos = y.detach().cpu()[:, :, ::2, ::2] > 0
        yr = F.interpolate(torch.where(pos, pre, pre * 0.1), scale_factor=2, mode="nearest")
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy.double())
        y.backward(gy.to(d))
        close(xd.grad, xr.grad, rel=5e-5, name="up_out gx")
        close(wd.grad, wr.grad, rel=5e-5, name="up_out gw")
        close(bd.grad, br.grad, rel=5e-5, name="up_out gb")


def test_conv_epilogue_residual_affine():
    """Inference epilogue: relu((conv + centre-crop(res)) * scale + shift)."""
    from spr_pick_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 30, 26, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) / 17
    res = torch.randn(2, 32, 34, 30, generator=g)
    scale, shift = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
    want = F.relu((F.conv2d(x.double(), w.double(), dilation=4) + res.double()[:, :, 6:-6, 6:-6])
                  * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    d = dev()
    xd, wd = x.to(d), w.to(d)
    geom = ops.make_geom(xd, None, wd, False, 1, 4, (0, 0, 0, 0))
    y = ops.conv2d_forward(xd, None, wd, geom, act=ops.ACT_RELU, scale=scale.to(d), shift=shift.to(d),
                           res=res.to(d), res_off=6)
    close(y, want, name="epilogue")


@pytest.mark.parametrize("shift", [1, 0])
@pytest.mark.parametrize("shape", [(3, 5, 64, 64), (2, 48, 8, 8), (4, 3, 2, 2), (1, 2, 6, 10)])
def test_shift_maxpool(shape, shift):
    from spr_pick_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(shape, generator=g)
    xr = x.double().requires_grad_(True)
    xs = F.pad(xr, (0, 0, shift, 0))[:, :, : shape[2]] if shift else xr
    yr = F.max_pool2d(xs, 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xd = x.to(dev()).requires_grad_(True)
    y = ops.shift_maxpool2(xd, shift)
    assert torch.equal(y.cpu(), yr.float())
    y.backward(gy.to(dev()))
    assert torch.equal(xd.grad.cpu(), xr.grad.float())


@pytest.mark.parametrize("shape", [(2, 3, 12, 4), (3, 48, 64, 64), (1, 2, 6, 10)])
def test_shift_maxpool_ties_go_to_the_first_maximum(shape):
    """Few distinct values: most windows have tied maxima; the gradient goes to the first one in row-major order
    (torch's rule), in the vectorised backward kernel (W % 4 == 0) and in the general one."""
    from spr_pick_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randint(0, 3, shape, generator=g).float()
    xr = x.double().requires_grad_(True)
    yr = F.max_pool2d(F.pad(xr, (0, 0, 1, 0))[:, :, : shape[2]], 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xd = x.to(dev()).requires_grad_(True)
    y = ops.shift_maxpool2(xd, 1)
    assert torch.equal(y.cpu(), yr.float())
    y.backward(gy.to(dev()))
    assert torch.equal(xd.grad.cpu(), xr.grad.float())


@pytest.mark.parametrize("B,C,P", [(2, 1, 64), (3, 2, 8), (1, 4, 2), (2, 3, 96), (1, 3, 128)])
def test_rot4_and_unrot4(B, C, P):
    from oracle import networks
    from spr_pick_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, C, P, P, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = torch.cat([networks.rot90cw(xr, a) for a in (0, 90, 180, 270)], 0)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.to(dev()).requires_grad_(True)
    y = ops.rot4_stack(xd)
    assert torch.equal(y.cpu(), yr.detach())
    y.backward(gy.to(dev()))
    close(xd.grad, xr.grad, rel=1e-6, name="rot4 grad")

    dd = torch.randn(4 * B, C, P, P, generator=g)
    dr = dd.clone().requires_grad_(True)
    s = networks.shift_down(dr, 1)
    fr = torch.cat([networks.rot90cw(q, a) for q, a in zip(torch.chunk(s, 4, 0), (0, 270, 180, 90))], 1)
    gf = torch.randn(fr.shape, generator=g)
    fr.backward(gf)
    d2 = dd.to(dev()).requires_grad_(True)
    f = ops.unrot4_shift_concat(d2)
    assert torch.equal(f.cpu(), fr.detach())
    f.backward(gf.to(dev()))
    assert torch.equal(d2.grad.cpu(), dr.grad)


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape", [(4, 32, 29, 29), (8, 1, 64, 64), (4, 128, 1, 1), (3, 64, 9, 9)])
def test_batch_norm(shape, relu):
    from spr_pick_amd import ops
    g = torch.Generator().manual_seed(3)
    C = shape[1]
    x = torch.randn(shape, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    yr = F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
    if relu:
        yr = F.relu(yr)
    gy = torch.randn(shape, generator=g)
    yr.backward(gy.double())
    d = dev()
    xd, gd, bd = (t.to(d).requires_grad_(True) for t in (x, gamma, beta))
    rm_d, rv_d = rm.to(d), rv.to(d)
    y = ops.batch_norm_train(xd, gd, bd, rm_d, rv_d, 0.1, 1e-5, relu)
    close(y, yr, rel=1e-5, name="bn y")
    y.backward(gy.to(d))
    close(rm_d, rm_r, rel=1e-5, name="running_mean")
    close(rv_d, rv_r, rel=1e-5, name="running_var")
    close(xd.grad, xr.grad, rel=1e-4, name="bn gx")
    close(gd.grad, gr.grad, rel=1e-4, name="bn ggamma")
    close(bd.grad, br.grad, rel=1e-4, name="bn gbeta")
    ye = ops.batch_norm_eval(x.to(d), gamma.to(d), beta.to(d), rm.to(d), rv.to(d), 1e-5, relu)
    want = F.batch_norm(x.double(), rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5)
    close(ye, F.relu(want) if relu else want, rel=1e-5, name="bn eval")


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape,groups", [((8, 32, 29, 29), 2), ((6, 64, 9, 9), 3), ((64, 1, 64, 64), 2), ((4, 128, 1, 1), 2)])
def test_batch_norm_groups(shape, groups, relu):
    """groups > 1 = the module called once per stacked pass (joint_network_v2.py runs the detector on the patches and on
    their flipped copies): per-pass batch statistics, running averages updated pass after pass."""
    from spr_pick_amd import ops
    g = torch.Generator().manual_seed(13)
    N, C = shape[0], shape[1]
    Ng = N // groups
    x = torch.randn(shape, generator=g) * 2 + torch.arange(N).view(N, 1, 1, 1) * 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    yr = torch.cat([F.batch_norm(xr[i * Ng:(i + 1) * Ng], rm_r, rv_r, gr, br, True, 0.1, 1e-5) for i in range(groups)], 0)
    if relu:
        yr = F.relu(yr)
    gy = torch.randn(shape, generator=g)
    yr.backward(gy.double())
    d = dev()
    xd, gd, bd = (t.to(d).requires_grad_(True) for t in (x, gamma, beta))
    rm_d, rv_d = rm.to(d), rv.to(d)
    y = ops.batch_norm_train(xd, gd, bd, rm_d, rv_d, 0.1, 1e-5, relu, groups=groups)
    close(y, yr, rel=1e-5, name="bn groups y")
    y.backward(gy.to(d))
    close(rm_d, rm_r, rel=1e-5, name="running_mean")
    close(rv_d, rv_r, rel=1e-5, name="running_var")
    close(xd.grad, xr.grad, rel=1e-4, name="bn groups gx")
    close(gd.grad, gr.grad, rel=1e-4, name="bn groups ggamma")
    close(bd.grad, br.grad, rel=1e-4, name="bn groups gbeta")
    with pytest.raises(Exception):
        ops.batch_norm_train(xd.detach()[:N - 1], gd, bd, rm_d, rv_d, 0.1, 1e-5, relu, groups=groups if (N - 1) % groups else groups + 1)


def test_reparam_sigmoid_ssdn():
    from oracle import pipeline
    from spr_pick_amd import ops
    g = torch.Generator().manual_seed(4)
    B, H, W = 3, 64, 64
    o = torch.randn(B, 2, H, W, generator=g)
    eps = torch.randn(B, 1, H, W, generator=g)
    x = torch.rand(B, 1, H, W, generator=g)
    ns = torch.rand(B, 1, 1, 1, generator=g) * 0.3 + 0.05
    d = dev()
    # reparameterize
    orr = o.double().requires_grad_(True)
    zr = orr[:, 0:1] + eps.double() * orr[:, 1:2] ** 2
    gz = torch.randn(zr.shape, generator=g)
    zr.backward(gz.double())
    od = o.to(d).requires_grad_(True)
    z = ops.reparameterize(od, eps.to(d))
    close(z, zr, rel=1e-6, name="z")
    z.backward(gz.to(d))
    close(od.grad, orr.grad, rel=1e-6, name="reparam grad")
    # sigmoid clamp (large logits exercise both clamp sides)
    lg = torch.randn(1000, generator=g) * 8
    lr_ = lg.double().requires_grad_(True)
    pr = torch.clamp(torch.sigmoid(lr_), 1e-4, 1 - 1e-4)
    gp = torch.randn(1000, generator=g)
    pr.backward(gp.double())
    ld = lg.to(d).requires_grad_(True)
    p = ops.sigmoid_clamp(ld)
    close(p, pr, rel=1e-6, name="sigmoid")
    p.backward(gp.to(d))
    close(ld.grad, lr_.grad, rel=1e-5, name="sigmoid grad")
    # ssdn
    orr = o.double().requires_grad_(True)
    nsr = ns.double().requires_grad_(True)
    nll, pme, sx, _ = pipeline.ssdn_terms(x.double(), orr[:, 0:1], orr[:, 1:2], nsr)
    lossr = nll.reshape(B, -1).mean(1, keepdim=True)
    gl = torch.randn(B, 1, generator=g)
    lossr.backward(gl.double())
    od = o.to(d).requires_grad_(True)
    nsd = ns.to(d).requires_grad_(True)
    loss, pm, ms, nsm = ops.ssdn_nll_pme(x.to(d), od, nsd)
    assert nsm.numel() == 0
    close(loss, lossr, rel=2e-6, name="ssdn loss")
    close(pm, pme, rel=2e-6, name="pme")
    close(ms, (sx ** 0.5)[:, 0].unsqueeze(0), rel=2e-6, name="model std")
    loss.backward(gl.to(d))
    close(od.grad, orr.grad, rel=1e-5, name="ssdn g_out")
    close(nsd.grad, nsr.grad, rel=1e-5, name="ssdn g_noise")
    # poisson likelihood (denoiser_v2.py:412-424): sigma_n^2 = max(mu, 1e-3) * estimate per pixel; mu on both sides of the floor
    op = o.clone()
    op[:, 0] = op[:, 0] * 0.5 + 0.1
    assert 0.15 < float((op[:, 0] < 1e-3).float().mean()) < 0.85
    orr = op.double().requires_grad_(True)
    nsr = ns.double().requires_grad_(True)
    nll, pme, sx, nstd = pipeline.ssdn_terms(x.double(), orr[:, 0:1], orr[:, 1:2], nsr, "poisson")
    lossr = nll.reshape(B, -1).mean(1, keepdim=True)
    lossr.backward(gl.double())
    od = op.to(d).requires_grad_(True)
    nsd = ns.to(d).requires_grad_(True)
    loss, pm, ms, nsm = ops.ssdn_nll_pme(x.to(d), od, nsd, ops.NOISE_POISSON)
    close(loss, lossr, rel=2e-6, name="poisson loss")
    close(pm, pme, rel=2e-6, name="poisson pme")
    close(nsm, nstd[:, 0], rel=2e-6, name="poisson noise std map")
    loss.backward(gl.to(d))
    close(od.grad, orr.grad, rel=1e-5, name="poisson g_out")
    close(nsd.grad, nsr.grad, rel=1e-5, name="poisson g_noise")


def test_cpu_tensors_are_refused():
    from spr_pick_amd import _lib, ops
    with pytest.raises(_lib.SprkError):
        ops.conv2d(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 3, 3))


# ---- 16-bit-operand MFMA convolutions (BASELINE configs[4]) ------------------------------------------------------
CONV16_CASES = [
    # name, N, C1, C2, H, W, Cout, K, pad(t,b,l,r), act, bias, up_out
    ("c16 enc1.0 shift 1->48 @64", 8, 1, 0, 64, 64, 48, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 enc1.2 shift 48->48 @64 (MT=4)", 32, 48, 0, 64, 64, 48, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 enc2 shift 48->48 @32", 16, 48, 0, 32, 32, 48, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 enc3 shift 48->48 @16", 64, 48, 0, 16, 16, 48, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 dec3.0 shift 96+48->96 @32", 16, 96, 48, 32, 32, 96, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 dec2.2 shift 96->96 @32 up_out", 8, 96, 0, 32, 32, 96, 3, (2, 0, 1, 1), 1, True, True),
    ("c16 dec1.0 shift 96+1->96 @64 (MT=4)", 32, 96, 1, 64, 64, 96, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 sigma plain 96+1->96 @64", 4, 96, 1, 64, 64, 96, 3, (1, 1, 1, 1), 1, True, False),
    ("c16 dec1.2 shift 96->96 @64", 8, 96, 0, 64, 64, 96, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 head 1x1 384->384 @64", 4, 384, 0, 64, 64, 384, 1, (0, 0, 0, 0), 1, True, False),
    ("c16 head kernel 1x1 384->384 @64 (all outputs per workgroup)", 8, 384, 0, 64, 64, 384, 1, (0, 0, 0, 0), 1, True, False),
    ("c16 head kernel 1x1 128->200 @20x16 (ragged pixels and channels)", 100, 128, 0, 20, 16, 200, 1, (0, 0, 0, 0), 2, False, False),
    ("c16 head 1x1 384->96 @64", 4, 384, 0, 64, 64, 96, 1, (0, 0, 0, 0), 1, True, False),
    ("c16 ragged 40->70 @24x48 pad(1,1,2,0)", 6, 40, 0, 24, 48, 70, 3, (1, 1, 2, 0), 2, False, False),
    ("c16 filled-size 48->48 @96x128", 1, 48, 0, 96, 128, 48, 3, (2, 0, 1, 1), 1, True, False),
    # large enough (>= 512 regions of 128 pixels) for the 16-bit backward-weight kernel as well
    ("c16 wgrad dec2.0 shift 96+48->96 @32", 64, 96, 48, 32, 32, 96, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 wgrad enc2 plain 48->48 @32", 64, 48, 0, 32, 32, 48, 3, (1, 1, 1, 1), 1, True, False),
    ("c16 wgrad 2 segments 96->96 @128x64", 4, 96, 0, 128, 64, 96, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 wgrad 40+30->88 @64x128 pad(1,1,2,0)", 8, 40, 30, 64, 128, 88, 3, (1, 1, 2, 0), 0, True, False),
    ("c16 wgrad dec3.0 shift 96+48->96 @16 (8 x 16 regions)", 256, 96, 48, 16, 16, 96, 3, (2, 0, 1, 1), 1, True, False),
    ("c16 wgrad enc3 plain 48->48 @16", 256, 48, 0, 16, 16, 48, 3, (1, 1, 1, 1), 1, True, False),
    # 1x1 backward-weight kernel: 192 x 192 blocks (Cout > 96), 96 x 384 blocks, ragged channel counts / region split
    ("c16 wgrad 1x1 384->384 @64", 16, 384, 0, 64, 64, 384, 1, (0, 0, 0, 0), 1, True, False),
    ("c16 wgrad 1x1 384->96 @64", 16, 384, 0, 64, 64, 96, 1, (0, 0, 0, 0), 1, True, False),
    ("c16 wgrad 1x1 200->130 @24x48 (ragged)", 58, 200, 0, 24, 48, 130, 1, (0, 0, 0, 0), 2, True, False),
    ("c16 wgrad 1x1 104->40 @32", 66, 104, 0, 32, 32, 40, 1, (0, 0, 0, 0), 0, False, False),
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", CONV16_CASES, ids=[c[0] for c in CONV16_CASES])
def test_conv2d_16bit_operands(case, dt):
    """Forward and backward-data on v_mfma_f32_16x16x32_{bf16,f16} (conv16.hip), two statements:
      (exact model)  the kernel computes conv(round16(x), round16(w)) with exact products and fp32 sums: against an
                     fp64 convolution of the operands rounded on the CPU (torch RNE casts) the budget is the fp32
                     path's own, 2e-5 / 5e-5 of the tensor scale — any wrong lane map, k order or missed conversion
                     fails this by orders of magnitude;
      (error bound)  against the UNROUNDED fp64 convolution every output is within 2u * sum_k |a_k| |w_k| + fp32
                     slack, u = 2^-8 (bf16) / 2^-11 (fp16): the bound DESIGN.md states for the 16-bit path.
    The launch counter proves the 16-bit kernel ran for forward and backward-data; the weight gradient is checked too
    (on whichever backward-weight kernel the library picks for the dtype)."""
    from spr_pick_amd import _lib, ops
    name, N, C1, C2, H, W, Cout, K, pad, act, has_b, up_out = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    x = torch.randn(N, C1, H, W, generator=g)
    x2 = torch.randn(N, C2, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, C1 + C2, K, K, generator=g) / np.sqrt((C1 + C2) * K * K)
    b = torch.randn(Cout, generator=g) * 0.1 if has_b else None
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    u = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11     # unit roundoff: 8 / 11 significant bits, round to nearest
    r16 = lambda t: None if t is None else t.to(tdt).double()
    d = dev()
    L = _lib.lib()
    dl = [t.to(d).requires_grad_(True) if t is not None else None for t in (x, x2, w, b)]
    n0, w0 = L.sprk_conv16_launch_count(), L.sprk_wgrad16_launch_count()
    y = ops.conv2d(dl[0], dl[2], dl[3], x2=dl[1], pad=pad, act=act, up_out=up_out, dtype=_lib.DTYPES[dt + "!"])
    assert L.sprk_conv16_launch_count() == n0 + 1, "forward did not take the 16-bit kernel"
    yv = y.detach().cpu()
    if up_out:
        assert torch.equal(yv[:, :, 0::2, 0::2], yv[:, :, 1::2, 1::2]) and torch.equal(yv[:, :, 0::2, 1::2], yv[:, :, 1::2, 0::2])
        yv = yv[:, :, 0::2, 0::2]
    bd = None if b is None else b.double()
    pre_r = ref_conv(r16(x), r16(x2), r16(w), bd, 0, 1, 1, pad, 0)
    act_f = (lambda t: F.leaky_relu(t, 0.1)) if act == 1 else (F.relu if act == 2 else (lambda t: t))
    close(yv, act_f(pre_r), name=name + " y (exact model)")
    pre_t = ref_conv(x.double(), None if x2 is None else x2.double(), w.double(), bd, 0, 1, 1, pad, 0)
    xin = x if x2 is None else torch.cat((x, x2), 1)
    mag = ref_conv(xin.double().abs(), None, w.double().abs(), None, 0, 1, 1, pad, 0)
    err = (yv.double() - act_f(pre_t)).abs()
    bound = 2 * u * mag + 2e-5 * pre_t.abs().max()
    assert bool((err <= bound).all()), "%s: rounding bound violated: worst ratio %.3f" % (name, float((err / bound).max()))

    # backward: gpre = gy * act'(y) is an fp32 product (act_bwd kernel); backward-data = conv16(round16(gpre), round16(w))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.to(d))
    torch.cuda.synchronize()
    # backward-data is a convolution over the layer's OUTPUT channels (k) producing its input channels (n): the
    # 16-bit kernel takes it when k is 8m or 8m+1 and there are >= 33 input channels
    bwd16 = Cout % 8 <= 1 and C1 + C2 >= 33
    assert L.sprk_conv16_launch_count() >= n0 + 1 + int(bwd16), "backward-data did not take the 16-bit kernel"
    gyl = gy
    if up_out:
        gyl = gy[:, :, 0::2, 0::2] + gy[:, :, 0::2, 1::2] + gy[:, :, 1::2, 0::2] + gy[:, :, 1::2, 1::2]
    if act:
        slope = torch.tensor(0.1 if act == 1 else 0.0)
        gpre = torch.where(yv > 0, gyl, gyl * slope)
    else:
        gpre = gyl
    pt, pb, pl, pr = pad
    # d/d(input) of a stride-1 correlation = full correlation of gpre with the flipped, channel-transposed taps
    wr = r16(w)
    gin = F.conv_transpose2d(r16(gpre), wr)[:, :, pt:pt + H, pl:pl + W]
    if bwd16 and not up_out:
        close(dl[0].grad, gin[:, :C1], rel=5e-5, name=name + " gx (exact model)")
        if x2 is not None:
            close(dl[1].grad, gin[:, C1:], rel=5e-5, name=name + " gx2 (exact model)")
    elif bwd16:
        # the 2x2 sum of the upsampled gradient is an fp32 sum whose order is the kernel's: a different order moves a
        # few gpre values across a 16-bit rounding boundary, so this case is held to the rounding budget instead
        close(dl[0].grad, gin[:, :C1], rel=8 * u, name=name + " gx (rounding budget)")
    # weight gradient: on the 16-bit backward-weight kernel (wgrad16.hip) when the layer is large enough — then it is
    # conv_weight(round16(x), round16(gpre)) with exact products and fp32 sums (exact model, 1e-4: sums of up to
    # 5e5 fp32 terms over 256 partial results) — otherwise on the fp32 kernels: the unrounded fp64 statement
    wg16 = (K == 3 and 33 <= Cout <= 96 and (W % 64 == 0 or W in (16, 32)) and H % (128 // min(W, 64)) == 0
            and N * H * W >= 65536 and not up_out)
    wg16 = wg16 or (K == 1 and Cout >= 33 and C1 >= 97 and C2 == 0 and (H * W) % 64 == 0 and N * H * W >= 65536)
    assert L.sprk_wgrad16_launch_count() == w0 + int(wg16), "backward-weight kernel choice"
    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (x, x2, w, b)]
    pre = ref_conv(leaves[0], leaves[1], leaves[2], leaves[3], 0, 1, 1, pad, 0)
    pre.backward(gpre.double())
    if wg16:
        xin = x if x2 is None else torch.cat((x, x2), 1)
        xp = F.pad(r16(xin), (pad[2], pad[3], pad[0], pad[1]))
        gw_model = torch.nn.grad.conv2d_weight(xp, w.shape, r16(gpre))
        close(dl[2].grad, gw_model, rel=1e-4, name=name + " gw (exact model)")
    close(dl[2].grad, leaves[2].grad, rel=8 * u, name=name + " gw")
    if b is not None:
        close(dl[3].grad, leaves[3].grad, rel=5e-5, name=name + " gb")


STORE16_CASES = [
    # name, N, C1, C2, H, W, Cout, K, pad, act, bias, up_out, x dtype ("16" | "32")
    ("s16 enc1.0 shift 1->48 @64 (fp32 image in, 16-bit out)", 32, 1, 0, 64, 64, 48, 3, (2, 0, 1, 1), 1, True, False, "32"),
    ("s16 enc1.2 shift 48->48 @64", 32, 48, 0, 64, 64, 48, 3, (2, 0, 1, 1), 1, True, False, "16"),
    ("s16 enc2 shift 48->48 @32", 64, 48, 0, 32, 32, 48, 3, (2, 0, 1, 1), 1, True, False, "16"),
    ("s16 dec2.0 shift 96+48->96 @32", 64, 96, 48, 32, 32, 96, 3, (2, 0, 1, 1), 1, True, False, "16"),
    ("s16 dec2.2 shift 96->96 @32 up_out", 64, 96, 0, 32, 32, 96, 3, (2, 0, 1, 1), 1, True, True, "16"),
    ("s16 dec1.0 shift 96+1->96 @64", 32, 96, 1, 64, 64, 96, 3, (2, 0, 1, 1), 1, True, False, "16"),
    ("s16 sigma plain 96->96 @64", 16, 96, 0, 64, 64, 96, 3, (1, 1, 1, 1), 1, True, False, "16"),
    ("s16 ragged 40+30->88 @64x128 pad(1,1,1,1)", 8, 40, 30, 64, 128, 88, 3, (1, 1, 1, 1), 0, True, False, "16"),
    ("s16 head 1x1 384->384 @64", 16, 384, 0, 64, 64, 384, 1, (0, 0, 0, 0), 1, True, False, "16"),
    ("s16 head 1x1 384->96 @64 (fp32 out as in front of output_conv)", 16, 384, 0, 64, 64, 96, 1, (0, 0, 0, 0), 1, True, False, "16"),
    ("s16 head 1x1 96->96 @64 (sigma net)", 16, 96, 0, 64, 64, 96, 1, (0, 0, 0, 0), 1, True, False, "16"),
    ("s16 head 1x1 128->200 @20x16 (ragged pixels and channels)", 100, 128, 0, 20, 16, 200, 1, (0, 0, 0, 0), 2, False, False, "16"),
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", STORE16_CASES, ids=[c[0] for c in STORE16_CASES])
def test_conv2d_16bit_storage(case, dt):
    """16-bit ACTIVATION TENSORS (SPRK_DT_X16 / SPRK_DT_Y16, round 4): the kernels read bf16 / fp16 tensors without a
    conversion and round the fp32 accumulator once, at the store.  Exact model: y = round16(act(conv(x16, round16(w)) + b))
    with exact products and fp32 sums — against the fp64 statement the budget is one 16-bit rounding of the result
    (u |y|) plus the fp32 path's own budget; the gradients likewise (gin = round16(conv^T(gpre16, round16(w))),
    gw = conv_weight(x16, gpre16) in fp32).  Tensors keep the type they were given; the launch counters prove the 16-bit
    kernels ran and no conversion pass was needed."""
    from spr_pick_amd import _lib, ops
    name, N, C1, C2, H, W, Cout, K, pad, act, has_b, up_out, xin_t = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    u = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    fp32_out = "fp32 out" in name
    x = torch.randn(N, C1, H, W, generator=g)
    x2 = torch.randn(N, C2, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, C1 + C2, K, K, generator=g) / np.sqrt((C1 + C2) * K * K)
    b = torch.randn(Cout, generator=g) * 0.1 if has_b else None
    d = dev()
    L = _lib.lib()
    # what the kernel is handed: 16-bit tensors (their values ARE the operands), or the fp32 image for the first layer
    x_in = x.to(tdt) if xin_t == "16" else x
    x2_in = None if x2 is None else (x2 if C2 == 1 else x2.to(tdt))       # the 1-channel raw image arrives in fp32 and is cast
    xl = x_in.to(d).requires_grad_(True)
    x2l = None if x2_in is None else x2_in.to(d).requires_grad_(C2 > 1)
    wl = w.to(d).requires_grad_(True)
    bl = None if b is None else b.to(d).requires_grad_(True)
    n0, w0 = L.sprk_conv16_launch_count(), L.sprk_wgrad16_launch_count()
    launches0 = L.sprk_launch_count()
    y = ops.conv2d(xl, wl, bl, x2=x2l, pad=pad, act=act, up_out=up_out, dtype=_lib.DTYPES[dt], store16=not fp32_out)
    assert L.sprk_conv16_launch_count() == n0 + 1, "forward did not take the 16-bit kernel"
    assert y.dtype == (torch.float32 if fp32_out else tdt)
    yv = y.detach().cpu()
    if up_out:
        assert torch.equal(yv[:, :, 0::2, 0::2], yv[:, :, 1::2, 1::2]) and torch.equal(yv[:, :, 0::2, 1::2], yv[:, :, 1::2, 0::2])
        yv = yv[:, :, 0::2, 0::2]
    r16 = lambda t: None if t is None else t.to(tdt).double()
    act_f = (lambda t: F.leaky_relu(t, 0.1)) if act == 1 else (F.relu if act == 2 else (lambda t: t))
    ref = act_f(ref_conv(r16(x), r16(x2), r16(w), None if b is None else b.double(), 0, 1, 1, pad, 0))
    err = (yv.double() - ref).abs()
    budget = (0.0 if fp32_out else u) * ref.abs() * 1.001 + 2e-5 * ref.abs().max()
    assert bool((err <= budget).all()), "%s: y beyond one rounding of the exact model: worst ratio %.3f" % (name, float((err / budget).max()))

    # backward: the gradient arrives in y's storage type
    gy = torch.randn(y.shape, generator=g).to(y.dtype)
    y.backward(gy.to(d))
    torch.cuda.synchronize()
    assert xl.grad.dtype == xl.dtype and (x2l is None or not x2l.requires_grad or x2l.grad.dtype == x2l.dtype)
    assert L.sprk_conv16_launch_count() >= n0 + (2 if C1 + C2 >= 33 else 1), "backward-data did not take the 16-bit kernel"
    gyl = gy.double()
    if up_out:
        gyl = (gyl[:, :, 0::2, 0::2] + gyl[:, :, 0::2, 1::2]) + (gyl[:, :, 1::2, 0::2] + gyl[:, :, 1::2, 1::2])
    slope = {0: 1.0, 1: 0.1, 2: 0.0}[act]
    gpre = torch.where(yv.double() > 0, gyl, gyl * slope) if act else gyl
    gpre16 = gpre.float().to(y.dtype)              # act_bwd stores the pre-activation gradient in the gradient's type
    pt, pb, pl, pr = pad
    if C1 + C2 >= 33:
        gin = F.conv_transpose2d(gpre16.double() if not fp32_out else r16(gpre16), r16(w))[:, :, pt:pt + H, pl:pl + W]
        gx = xl.grad.detach().cpu().double()
        e = (gx - gin[:, :C1]).abs()
        # (an up-sampled gradient is 2x2-summed in fp32 in the kernel's order before it is rounded: rounding budget there)
        bud = (8 * u if up_out else (u if xl.dtype == tdt else 0.0)) * gin[:, :C1].abs() * 1.001 + (8 * u if up_out else 5e-5) * gin.abs().max()
        assert bool((e <= bud).all()), "%s: gx beyond the budget: worst ratio %.3f" % (name, float((e / bud).max()))
        if x2l is not None and x2l.requires_grad:
            e2 = (x2l.grad.detach().cpu().double() - gin[:, C1:]).abs()
            assert bool((e2 <= u * gin[:, C1:].abs() * 1.001 + 5e-5 * gin.abs().max()).all()), name + " gx2"
    # weight gradient
    wg16 = (K == 3 and 33 <= Cout <= 96 and (W % 64 == 0 or W in (16, 32)) and H % (128 // min(W, 64)) == 0
            and N * H * W >= 65536)
    wg16 = wg16 or (K == 1 and Cout >= 33 and C1 >= 97 and C2 == 0 and (H * W) % 64 == 0 and N * H * W >= 65536)
    assert L.sprk_wgrad16_launch_count() == w0 + int(wg16), "backward-weight kernel choice"
    xin = r16(x) if x2 is None else torch.cat((r16(x), r16(x2)), 1)
    xp = F.pad(xin, (pad[2], pad[3], pad[0], pad[1]))
    gw_model = torch.nn.grad.conv2d_weight(xp, w.shape, r16(gpre16))
    close(wl.grad, gw_model, rel=1e-4 if wg16 else 8 * u, name=name + " gw (exact model)")
    if b is not None:
        # (the bias gradient sums the fp32 products gy * act'(y) BEFORE they are rounded for storage)
        close(bl.grad, gpre.sum((0, 2, 3)), rel=5e-5, name=name + " gb")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_plumbing_kernels_on_16bit_tensors(dt):
    """Pooling, un-rotation and the activation backward with 16-bit tensors: pooling and un-rotation move values (max
    commutes with a monotone rounding), so they equal the fp32 kernels on the same values bit for bit; the activation
    backward computes in fp32 and rounds once, in every combination of input / output storage types."""
    from spr_pick_amd import _lib, ops
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    d = dev()
    g = torch.Generator().manual_seed(77)
    for shift in (0, 1):
        x = torch.randn(6, 5, 32, 48, generator=g).to(tdt)
        x[0, 0, 4:6, 8:10] = 1.5                                    # a tie: the first maximum takes the gradient
        x16 = x.to(d).requires_grad_(True)
        x32 = x.float().to(d).requires_grad_(True)
        for fuse_act in (ops.ACT_NONE, ops.ACT_LEAKY):
            y16, y32 = ops.shift_maxpool2(x16, shift, fuse_act), ops.shift_maxpool2(x32, shift, fuse_act)
            assert y16.dtype == tdt and torch.equal(y16.float(), y32)
            gy = torch.randn(y32.shape, generator=g).to(tdt)
            g16, = torch.autograd.grad(y16, x16, gy.to(d))
            g32, = torch.autograd.grad(y32, x32, gy.float().to(d))
            assert g16.dtype == tdt and torch.equal(g16.float(), g32.to(tdt).float())
    for P in (32, 64, 128):           # the 32x32-tile kernel, the 64x64-tile kernel with one and with four tiles per plane
        dd = torch.randn(8, 6, P, P, generator=g).to(tdt)
        d16, d32 = dd.to(d).requires_grad_(True), dd.float().to(d).requires_grad_(True)
        f16, f32 = ops.unrot4_shift_concat(d16), ops.unrot4_shift_concat(d32)
        assert f16.dtype == tdt and torch.equal(f16.float(), f32)
        gf = torch.randn(f32.shape, generator=g).to(tdt)
        a16, = torch.autograd.grad(f16, d16, gf.to(d))
        a32, = torch.autograd.grad(f32, d32, gf.float().to(d))
        assert a16.dtype == tdt and torch.equal(a16.float(), a32)
    # activation backward: gpre = gy * act'(y), bias sums in fp32; every storage combination, dense / upsampled / sliced gy
    S = torch.ops.sprk
    N, C, H, W = 6, 10, 16, 24
    y = torch.randn(N, C, H, W, generator=g)
    gy = torch.randn(N, C, H, W, generator=g)
    for tg in (torch.float32, tdt):
        for ty in (torch.float32, tdt):
            for to in (torch.float32, tdt):
                gyt, yt = gy.to(tg), y.to(ty)
                gb = torch.zeros(C, device=d)
                out = S.act_bwd(gyt.to(d), yt.to(d), ops.ACT_LEAKY, [N, C, H, W], 0, True, gb, False, {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[to])
                want = torch.where(yt.float() > 0, gyt.float(), gyt.float() * 0.1)
                assert out.dtype == to and torch.equal(out.cpu(), want.to(to)), (tg, ty, to)
                close(gb, want.double().sum((0, 2, 3)), rel=1e-5, name="bias sum")
    yu = torch.randn(N, C, 2 * H, 2 * W, generator=g).to(tdt)
    yu = yu[:, :, 0::2, 0::2].repeat_interleave(2, 2).repeat_interleave(2, 3).contiguous()
    gu = torch.randn(N, C + 4, 2 * H, 2 * W, generator=g).to(tdt)
    out = S.act_bwd(gu.to(d)[:, 2:2 + C], yu.to(d), ops.ACT_LEAKY, [N, C, H, W], 1, True, None, False, 1 if dt == "bf16" else 2)
    gs = gu[:, 2:2 + C].float()
    sm = (gs[:, :, 0::2, 0::2] + gs[:, :, 0::2, 1::2]) + (gs[:, :, 1::2, 0::2] + gs[:, :, 1::2, 1::2])
    want = torch.where(yu[:, :, 0::2, 0::2].float() > 0, sm, sm * 0.1).to(tdt)
    assert torch.equal(out.cpu(), want)


def test_reduce_items_kinds():
    """sprk_reduce_items: the three layouts, 50 items (two launches), odd part counts, against fp64 sums and against
    the documented order (four interleaved chains)."""
    import ctypes
    from spr_pick_amd import _lib, ops
    d = dev()
    L = _lib.lib()
    g = torch.Generator().manual_seed(21)
    items, keep, want = [], [], []

    def chains(t, dim):          # (s0 + s1) + (s2 + s3), s_j = sequential fp32 sum over p = j mod 4
        parts = t.shape[dim]
        s = []
        for j in range(4):
            acc = torch.zeros_like(t.select(dim, 0))
            for p in range(j, parts, 4):
                acc = acc + t.select(dim, p)
            s.append(acc)
        return (s[0] + s[1]) + (s[2] + s[3])

    for i in range(50):
        kind = 1 + i % 3
        parts = [1, 2, 3, 5, 8, 13, 64, 37][i % 8]
        if kind == 1:
            n = 100 + 37 * i
            src = torch.randn(parts, n, generator=g)
            ref = chains(src, 0)
            it = _lib.ReduceItem(None, None, 1, parts, n, 0, 0, 0)
        elif kind == 2:
            n = 5 + i
            src = torch.randn(n, parts, generator=g)
            ref = chains(src, 1)
            it = _lib.ReduceItem(None, None, 2, parts, n, 0, 0, 0)
        else:
            K, Cout, CoutP = 9 + i, 20 + i % 7, 32
            src = torch.randn(parts, K, CoutP, generator=g)
            ref = chains(src, 0)[:, :Cout].t().contiguous()     # [Cout][K]
            it = _lib.ReduceItem(None, None, 3, parts, 0, K, Cout, CoutP)
        sd = src.to(d)
        dst = torch.full(ref.shape, float("nan"), device=d)
        it.src, it.dst = sd.data_ptr(), dst.data_ptr()
        items.append(it); keep.append((sd, dst)); want.append(ref)
    arr = (_lib.ReduceItem * len(items))(*items)
    before = L.sprk_launch_count()
    _lib.check(L.sprk_reduce_items(arr, len(items), ops._stream(keep[0][0])), "sprk_reduce_items")
    assert L.sprk_launch_count() - before == 2            # 48 + 2 items
    for (sd, dst), ref in zip(keep, want):
        assert torch.equal(dst.cpu(), ref), "reduce_items: order of summation differs from the documented one"
    bad = (_lib.ReduceItem * 1)(_lib.ReduceItem(keep[0][0].data_ptr(), keep[0][1].data_ptr(), 7, 1, 1, 0, 0, 0))
    assert L.sprk_reduce_items(bad, 1, ops._stream(keep[0][0])) != 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_deferred_reductions_match_immediate(dtype):
    """Weight / bias gradients written through graph_step.FlatGrads (second-stage sums pending until the context
    closes, one launch) are bit-identical to the plain operators; a parameter with two consumers is finished early."""
    from spr_pick_amd import _lib, graph_step, ops, torch_ops
    d = dev()
    g = torch.Generator().manual_seed(22)
    dt = _lib.DTYPES[dtype]
    x = torch.randn(32, 48, 64, 64, generator=g).to(d)
    shapes = [(48, 48, 3), (96, 48, 3), (96, 96, 1), (8, 96, 3)]
    params = []
    for co, ci, k in shapes:
        params.append(torch.nn.Parameter((torch.randn(co, ci, k, k, generator=g) * 0.05).to(d)))
        params.append(torch.nn.Parameter(torch.randn(co, generator=g).to(d)))

    def net():
        h = x
        for i, (co, ci, k) in enumerate(shapes):
            pad = (2, 0, 1, 1) if k == 3 else (0, 0, 0, 0)
            h = ops.conv2d(h, params[2 * i], params[2 * i + 1], pad=pad, act=ops.ACT_LEAKY, dtype=dt)
        # the first layer's weights once more: a second consumer of the same parameter
        h2 = ops.conv2d(x, params[0], params[1], pad=(2, 0, 1, 1), act=ops.ACT_LEAKY, dtype=dt)
        return h.square().mean() + h2.mean()

    net().backward()
    want = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    fg = graph_step.FlatGrads(params)
    fg.begin_step()
    before = _lib.lib().sprk_launch_count()
    with fg:
        loss = net()
        loss.backward()
        fg.adopt_strays()
    assert torch_ops.pending_count(d) == 0
    for p, w in zip(params, want):
        assert torch.equal(p.grad, w), "deferred reduction changed a gradient (%s)" % (tuple(p.shape),)
    fg.check_adopted()
    # without the shared parameter every sum stays pending until the context closes
    for p in params:
        p.grad = None
    fg.begin_step()
    with fg:
        h = ops.conv2d(x, params[0], params[1], pad=(2, 0, 1, 1), act=ops.ACT_LEAKY, dtype=dt)
        h = ops.conv2d(h, params[2], params[3], pad=(2, 0, 1, 1), act=ops.ACT_LEAKY, dtype=dt)
        h.square().mean().backward()
        # two weight + two bias sums; the 48 -> 96 layer's weight gradient takes the Winograd backward-weight kernel at this
        # size (2048 regions, fp32), which finishes its own partial sums: 3 pending then
        assert torch_ops.pending_count(d) == (3 if dtype == "f32" else 4)
    assert torch_ops.pending_count(d) == 0


def test_pu_loss_kernel_every_label_mix():
    """ops.pu_loss (one launch: value + gradient) against the oracle's statement of utils/losses.py:303-349 and the
    gradient autograd derives from the torch mask form, for mixed, all-unlabelled, all-labelled, single-unlabelled
    batches and a batch longer than the workgroup."""
    from oracle import pipeline
    from spr_pick_amd.denoiser import PuLoss
    d = dev()
    g = torch.Generator().manual_seed(23)
    pu = PuLoss()
    for B, labels in ((16, None), (16, "none"), (8, "all"), (8, "one_unl"), (64, None), (700, None)):
        p = torch.rand(B, 1, 1, 1, generator=g) * 0.98 + 0.01
        y = torch.where(torch.rand(B, 1, generator=g) < 0.3, torch.rand(B, 1, generator=g), torch.full((B, 1), -1.0))
        if labels == "none":
            y = torch.full((B, 1), -1.0)
        elif labels == "all":
            y = torch.rand(B, 1, generator=g)
        elif labels == "one_unl":
            y = torch.rand(B, 1, generator=g)
            y[3] = -1.0
        for tau in (0.01, 0.2):
            pr = p.double().requires_grad_(True)
            want = pu.mask_form(tau, pr, y.double())
            want.backward()
            oracle_val = pipeline.pu_loss(tau, p, y)
            pd = p.to(d).requires_grad_(True)
            got = pu(tau, pd, y.to(d))
            (got * 1.5).backward()
            assert abs(float(got.detach()) - float(oracle_val)) <= 2e-5 * abs(float(oracle_val)) + 1e-5, (B, labels, tau)
            close(got, want.float(), rel=2e-5, name="pu loss %s %s" % (B, labels))
            close(pd.grad, 1.5 * pr.grad.float(), rel=2e-4, name="pu grad %s %s %s" % (B, labels, tau))
    with pytest.raises(Exception):
        pu(0.01, torch.rand(4), torch.zeros(4))       # CPU tensors: the product path is the GPU kernel


@pytest.mark.parametrize("up_out", [False, True])
def test_concat_halves_are_views_and_act_bwd_reads_them_strided(up_out):
    """The input gradient of a concat layer is handed on as two channel-slice VIEWS (no copy kernel); the producing
    layers' act_bwd reads such a gradient in place (dense planes, strided images), also in its 2x2-summing form."""
    from spr_pick_amd import ops
    g = torch.Generator().manual_seed(41)
    d = dev()
    N, H, W = 6, 16, 32
    a = torch.randn(N, 5, H, W, generator=g)
    b = torch.randn(N, 7, 2 * H, 2 * W, generator=g)
    w1 = torch.randn(24, 5, 3, 3, generator=g) * 0.2
    w2 = torch.randn(9, 7, 3, 3, generator=g) * 0.2
    w3 = torch.randn(16, 33, 3, 3, generator=g) * 0.1
    b1, b2, b3 = torch.randn(24, generator=g), torch.randn(9, generator=g), torch.randn(16, generator=g)
    if not up_out:
        a = torch.randn(N, 5, 2 * H, 2 * W, generator=g)

    def ref():
        t = [x.double().requires_grad_(True) for x in (w1, b1, w2, b2, w3, b3)]
        y1 = F.leaky_relu(F.conv2d(a.double(), t[0], t[1], padding=1), 0.1)
        if up_out:
            y1 = F.interpolate(y1, scale_factor=2, mode="nearest")
        y2 = F.relu(F.conv2d(b.double(), t[2], t[3], padding=1))
        z = F.leaky_relu(F.conv2d(torch.cat((y1, y2), 1), t[4], t[5], padding=1), 0.1)
        (z * z).sum().backward()
        return [x.grad for x in t]

    t = [x.to(d).requires_grad_(True) for x in (w1, b1, w2, b2, w3, b3)]
    y1 = ops.conv2d(a.to(d), t[0], t[1], pad=(1, 1, 1, 1), act=1, up_out=up_out)
    y2 = ops.conv2d(b.to(d), t[2], t[3], pad=(1, 1, 1, 1), act=2)
    seen = []
    y1.register_hook(lambda gr: seen.append(("y1", gr.is_contiguous(), gr.stride(0))))
    y2.register_hook(lambda gr: seen.append(("y2", gr.is_contiguous(), gr.stride(0))))
    z = ops.conv2d(y1, t[4], t[5], x2=y2, pad=(1, 1, 1, 1), act=1)
    (z * z).sum().backward()
    plane = 4 * H * W
    assert sorted(seen) == [("y1", False, 33 * plane), ("y2", False, 33 * plane)], seen     # views of the 33-channel gin
    for got, want, name in zip([x.grad for x in t], ref(), ("w1", "b1", "w2", "b2", "w3", "b3")):
        close(got, want, rel=5e-5, name="strided gy " + name)


# ---- activation backward fused into the neighbouring operators (ops.conv2d: premasked / x_act) ----------------------
CHAIN_CASES = [
    # name, N, Cin, C2(skip of the first conv), Cmid, Cout, H, W, shift-conv?, up_out of the second conv
    ("dec1 96+1->96->96 @64 (Winograd epilogue mask)", 4, 96, 1, 96, 96, 64, 64, True, False),
    ("dec2 96+48->96->96 @32 up_out", 8, 96, 48, 96, 96, 32, 32, True, True),
    ("enc1 1->48->48 @64", 8, 1, 0, 48, 48, 64, 64, True, False),
    ("sigma 96+1->96->96 @64 plain", 2, 96, 1, 96, 96, 64, 64, False, False),
    ("small plane 96->96->96 @8 (in-place mask after the direct kernel)", 8, 96, 0, 96, 96, 8, 8, True, False),
    ("odd 5->7->6 @12x20", 3, 5, 0, 7, 6, 12, 20, True, False),
]


@pytest.mark.parametrize("case", CHAIN_CASES, ids=[c[0] for c in CHAIN_CASES])
def test_conv_chain_with_fused_activation_backward(case):
    """conv -> LeakyReLU -> conv with the first layer's activation backward running inside the second layer's
    backward-data kernel (sprk_conv2d_bwd_data_masked): every gradient equals the fp64 autograd of the unfused maths
    at the operator budgets, and is bit-identical to the unfused HIP path where the mask is a pure multiplication by
    1 or 0.1 of the same value."""
    from spr_pick_amd import ops
    _, N, Cin, C2, Cmid, Cout, H, W, shifted, up_out = case
    g = torch.Generator().manual_seed(zlib.crc32(case[0].encode()))
    x = torch.randn(N, Cin, H, W, generator=g)
    x2 = torch.randn(N, C2, H, W, generator=g) if C2 else None
    w1 = torch.randn(Cmid, Cin + C2, 3, 3, generator=g) * (1.0 / (3 * (Cin + C2) ** 0.5))
    b1 = torch.randn(Cmid, generator=g) * 0.1
    w2 = torch.randn(Cout, Cmid, 3, 3, generator=g) * (1.0 / (3 * Cmid ** 0.5))
    b2 = torch.randn(Cout, generator=g) * 0.1
    pad = (2, 0, 1, 1) if shifted else (1, 1, 1, 1)
    # fp64 statement
    leaves = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)] + ([x2.double().requires_grad_(True)] if C2 else [])
    xr, w1r, b1r, w2r, b2r = leaves[:5]
    h = ref_conv(xr, leaves[5] if C2 else None, w1r, b1r, 0, 1, 1, pad, 1)
    yr = ref_conv(h, None, w2r, b2r, 0, 1, 1, pad, 1)
    if up_out:
        yr = F.interpolate(yr, scale_factor=2, mode="nearest")
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())

    def run(fuse):
        d = dev()
        t = [v.to(d).requires_grad_(True) for v in (x, w1, b1, w2, b2)] + ([x2.to(d).requires_grad_(True)] if C2 else [])
        hh = ops.conv2d(t[0], t[1], t[2], x2=t[5] if C2 else None, pad=pad, act=ops.ACT_LEAKY, premasked=fuse)
        yy = ops.conv2d(hh, t[3], t[4], pad=pad, act=ops.ACT_LEAKY, up_out=up_out, x_act=ops.ACT_LEAKY if fuse else ops.ACT_NONE)
        yy.backward(gy.to(d))
        return yy, t

    y1, fused = run(True)
    y0, plain = run(False)
    close(y1, yr, name="y")
    names = ["x", "w1", "b1", "w2", "b2", "x2"]
    for i, (a, b, r) in enumerate(zip(fused, plain, leaves)):
        # same kernels, the mask applied at another place: identical values (x * 1 or x * 0.1f either way)
        assert torch.equal(a.grad, b.grad), "d %s: fused and unfused paths differ by %.3e" % (
            names[i], float((a.grad - b.grad).abs().max()))
        # against fp64: a hidden activation within fp32 rounding of 0 takes slope 1 here and 0.1 there (or vice versa),
        # which moves the gradients that pass through that one pixel: at most 0.5 % of the elements may be beyond the
        # operator budget, none beyond 1 % of the tensor's scale (the unfused path shows the same elements: checked
        # bit for bit above)
        got, want = a.grad.detach().cpu().double(), r.grad.detach().double()
        scale = float(want.abs().max()) + 1e-30
        err = (got - want).abs()
        budget = (5e-5 if names[i] in ("x", "x2") else 1e-4) * scale
        assert float(err.max()) <= 1e-2 * scale, "d %s: max err %.3e of %.3e" % (names[i], float(err.max()), scale)
        assert int((err > budget).sum()) <= max(1, err.numel() // 200), "d %s: %d of %d beyond the budget" % (
            names[i], int((err > budget).sum()), err.numel())


@pytest.mark.parametrize("shift", [1, 0])
@pytest.mark.parametrize("shape", [(4, 48, 64, 64), (2, 5, 6, 10), (3, 48, 8, 8)])
def test_conv_pool_with_fused_activation_backward(shape, shift):
    """conv -> LeakyReLU -> (shifted) max-pool with the activation backward inside the pooling backward kernel."""
    from spr_pick_amd import ops
    N, C, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, 7, H, W, generator=g)
    w = torch.randn(C, 7, 3, 3, generator=g) * 0.2
    b = torch.randn(C, generator=g) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    h = ref_conv(xr, None, wr, br, 0, 1, 1, (2, 0, 1, 1), 1)
    hs = F.pad(h, (0, 0, shift, 0))[:, :, :H] if shift else h
    yr = F.max_pool2d(hs, 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    res = []
    for fuse in (True, False):
        d = dev()
        t = [v.to(d).requires_grad_(True) for v in (x, w, b)]
        hh = ops.conv2d(t[0], t[1], t[2], pad=(2, 0, 1, 1), act=ops.ACT_LEAKY, premasked=fuse)
        yy = ops.shift_maxpool2(hh, shift, ops.ACT_LEAKY if fuse else ops.ACT_NONE)
        yy.backward(gy.to(d))
        res.append(t)
        close(yy, yr, name="pooled")
    for i, n in enumerate(("x", "w", "b")):
        assert torch.equal(res[0][i].grad, res[1][i].grad), n
        got, want = res[0][i].grad.detach().cpu().double(), (xr, wr, br)[i].grad.detach().double()
        scale = float(want.abs().max()) + 1e-30
        err = (got - want).abs()
        assert float(err.max()) <= 1e-2 * scale and int((err > 1e-4 * scale).sum()) <= max(1, err.numel() // 200), n


PREP_CASES = ["enc1.2 shift 48->48 @64", "dec1.0 shift up(96)+1->96 @64", "dec1.2 shift 96->96 @64", "enc4 shift 48->48 @8",
              "head 1x1 384->384 @64", "head 1x1 96->2 @64", "det 3x3 d2 s2 32->64 @21", "det conv7 s2 1->32 @64"]


@pytest.mark.parametrize("dt", ["f32", "bf16!"])
@pytest.mark.parametrize("name", PREP_CASES)
def test_prepared_weights_are_bit_identical_to_per_call_transforms(name, dt):
    """ops.WeightPrep / SPRK_DT_WPREP: after the weights change, ONE sprk_prepare_weights launch re-runs the recorded
    transforms and the convolution calls skip their own — outputs and every gradient bit-identical to the plain calls
    (direct implicit GEMM, Winograd and 16-bit kernels; forward and backward-data)."""
    from spr_pick_amd import _lib, ops
    case = next(c for c in CONV_CASES if c[0] == name)
    _, N, C1, C2, H, W, up1, Cout, K, stride, dil, pad, act, has_bias = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    h1, w1 = (H // 2, W // 2) if up1 else (H, W)
    d = dev()
    x = torch.randn(N, C1, h1, w1, generator=g).to(d)
    x2 = torch.randn(N, C2, H, W, generator=g).to(d) if C2 else None
    w = (torch.randn(Cout, C1 + C2, K, K, generator=g) * 0.1).to(d)
    b = (torch.randn(Cout, generator=g) * 0.1).to(d) if has_bias else None
    code = _lib.DTYPES[dt]

    def run(wv):
        xs = x.clone().requires_grad_(True)
        ws_ = wv.clone().requires_grad_(True)       # (a fresh tensor: plain path)
        y = ops.conv2d(xs, ws_, b, x2=x2, up1=bool(up1), stride=stride, dil=dil, pad=pad, act=act, dtype=code)
        y.sum().backward()
        return y.detach(), xs.grad

    prep = ops.WeightPrep()
    wp = w.clone().requires_grad_(True)             # the persistent parameter of the prepared path

    def run_prep():
        xs = x.clone().requires_grad_(True)
        with prep:
            prep.begin_step(x)
            y = ops.conv2d(xs, wp, b, x2=x2, up1=bool(up1), stride=stride, dil=dil, pad=pad, act=act, dtype=code)
            y.sum().backward()
        return y.detach(), xs.grad

    y0, gx0 = run(w)
    y1, gx1 = run_prep()                            # records, transforms inside the calls
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
    n_items = len(prep.items)
    with torch.no_grad():
        wp.mul_(1.25).add_(0.01)                    # "the optimiser ran"
    y2, gx2 = run_prep()                            # one launch for all items, calls skip their transform
    y3, gx3 = run(wp.detach())
    assert prep.launches == (1 if n_items else 0) and len(prep.items) == n_items
    assert torch.equal(y2, y3) and torch.equal(gx2, gx3), (float((y2 - y3).abs().max()), float((gx2 - gx3).abs().max()))
    assert not torch.equal(y2, y1)


@pytest.mark.parametrize("shape", [(2, 384, 64, 64, 2), (1, 384, 32, 96, 2), (3, 96, 16, 40, 1), (1, 96, 128, 128, 1),
                                   (5, 384, 8, 16, 2)])
def test_fused_head_matches_three_convolutions(shape):
    """sprk_head1x1_fwd (one launch: 384 -> 384 -> 96 -> 2 or 96 -> 96 -> 96 -> 1 with LeakyReLU(0.1), the per-pixel head
    of the U-Nets, joint_network_v2.py:123-153, 241-244) against the fp64 statement at the operator budget, and against
    the three-launch HIP path."""
    from spr_pick_amd import networks, ops
    B, K0, H, W, N3 = shape
    g = torch.Generator().manual_seed(K0 + H + N3)
    f = torch.randn(B, K0, H, W, generator=g)
    c1 = networks.Conv2d(K0, K0, 1, act=ops.ACT_LEAKY)
    c2 = networks.Conv2d(K0, 96, 1, act=ops.ACT_LEAKY)
    c3 = networks.Conv2d(96, N3, 1, act=ops.ACT_NONE)
    with torch.no_grad():
        for c in (c1, c2, c3):
            c.weight.copy_(torch.randn(c.weight.shape, generator=g) * (1.5 / c.weight.shape[1] ** 0.5))
            c.bias.copy_(torch.randn(c.bias.shape, generator=g) * 0.3)
    fd = f.double()
    h = F.leaky_relu(F.conv2d(fd, c1.weight.double(), c1.bias.double()), 0.1)
    h = F.leaky_relu(F.conv2d(h, c2.weight.double(), c2.bias.double()), 0.1)
    want = F.conv2d(h, c3.weight.double(), c3.bias.double())
    d = dev()
    for c in (c1, c2, c3):
        c.to(d)
    with torch.no_grad():
        x = f.to(d)
        assert ops.head1x1_eligible(x, c1, c2, c3)
        got = ops.head1x1(x, c1, c2, c3)
        three = c3(c2(c1(x)))
    close(got, want, name="fused head")
    close(got, three, name="fused vs three launches")
    assert not ops.head1x1_eligible(x.requires_grad_(True), c1, c2, c3) or not torch.is_grad_enabled()


@pytest.mark.parametrize("B,P,N3", [(2, 64, 2), (1, 32, 2), (1, 16, 1), (3, 96, 2)])
def test_fused_head_on_the_rotated_stack_is_bit_identical_to_unrot_plus_head(B, P, N3):
    """sprk_head1x1_unrot_fwd: Shift2d + chunk + rotate + concat (joint_network_v2.py:230-239) as the input gather of the
    fused head — bit-identical to sprk_unrot4_shift_concat_fwd followed by sprk_head1x1_fwd (same values, same order),
    which the previous test pins against fp64."""
    from spr_pick_amd import networks, ops
    g = torch.Generator().manual_seed(B * 100 + P)
    dd = torch.randn(4 * B, 96, P, P, generator=g)
    c1 = networks.Conv2d(384, 384, 1, act=ops.ACT_LEAKY)
    c2 = networks.Conv2d(384, 96, 1, act=ops.ACT_LEAKY)
    c3 = networks.Conv2d(96, N3, 1, act=ops.ACT_NONE)
    with torch.no_grad():
        for c in (c1, c2, c3):
            c.weight.copy_(torch.randn(c.weight.shape, generator=g) * (1.5 / c.weight.shape[1] ** 0.5))
            c.bias.copy_(torch.randn(c.bias.shape, generator=g) * 0.3)
    d = dev()
    for c in (c1, c2, c3):
        c.to(d)
    with torch.no_grad():
        x = dd.to(d)
        assert ops.head1x1_unrot_eligible(x, c1, c2, c3)
        got = ops.head1x1_unrot(x, c1, c2, c3)
        f = ops.unrot4_shift_concat(x)
        if ops.head1x1_eligible(f, c1, c2, c3):
            assert torch.equal(got, ops.head1x1(f, c1, c2, c3))
        close(got, c3(c2(c1(f))), name="vs three launches")
    assert tuple(got.shape) == (B, N3, P, P)


@pytest.mark.parametrize("off,stride,with_y", [(3, 1, True), (6, 2, True), (3, 1, False), (1, 2, False)])
def test_crop_add_matches_slicing(off, stride, with_y):
    """ResidA's residual (models/feature_extractor.py:403-411) as one launch each way: values and both gradients equal
    autograd's slicing exactly (the op only moves and adds numbers)."""
    from spr_pick_amd import ops
    torch.manual_seed(off * 10 + stride)
    x = torch.randn(3, 5, 23, 21, device="cuda", requires_grad=True)
    xr = x.detach().clone().requires_grad_(True)
    ref_crop = xr[:, :, off:-off, off:-off][:, :, ::stride, ::stride]
    y = torch.randn_like(ref_crop).requires_grad_(True) if with_y else None
    yr = y.detach().clone().requires_grad_(True) if with_y else None
    out = ops.crop_add(y, x, off, stride)
    ref = ref_crop + yr if with_y else ref_crop
    assert out.shape == ref.shape and torch.equal(out, ref)
    g = torch.randn_like(out)
    out.backward(g)
    ref.backward(g)
    assert torch.equal(x.grad, xr.grad)
    if with_y:
        assert torch.equal(y.grad, yr.grad)


def test_noise_std_from_map_matches_torch():
    """softplus(mean(est) - 4) + 1e-3 (denoiser_v2.py:392-402), incl. the threshold-20 branch and very negative means."""
    from spr_pick_amd import ops
    torch.manual_seed(5)
    est = torch.randn(7, 1, 64, 64, device="cuda") * 0.3
    est += torch.tensor([0.0, 2.0, 4.0, 8.0, 30.0, -20.0, 3.9], device="cuda").view(-1, 1, 1, 1)
    a, b = est.clone().requires_grad_(True), est.clone().double().requires_grad_(True)
    out = ops.noise_std_from_map(a)
    ref = F.softplus(torch.mean(b, dim=(2, 3), keepdim=True) - 4.0) + 1e-3
    assert out.shape == (7, 1, 1, 1)
    assert torch.allclose(out.double(), ref, rtol=2e-6, atol=1e-9), float((out.double() - ref).abs().max())
    g = torch.randn_like(out)
    out.backward(g)
    ref.backward(g.double())
    assert torch.allclose(a.grad.double(), b.grad, rtol=2e-6, atol=1e-12), float((a.grad.double() - b.grad).abs().max())


@pytest.mark.parametrize("axis", [-1, -2])
@pytest.mark.parametrize("shape", [(32, 1, 1, 1), (5, 1, 3, 4)])
def test_joint_loss_matches_torch(axis, shape):
    """final = alpha * loss_out + (1 - alpha) * pred + 0.1 * mse(p, flip(pf)) (denoiser_v2.py:516-519) and its four
    gradients against autograd on the reference's own formula, in fp64."""
    from spr_pick_amd import ops
    torch.manual_seed(11)
    B, alpha = shape[0], 0.75
    lo = torch.rand(B, 1, device="cuda") * 3
    pred = torch.rand((), device="cuda") * 2
    p, pf = torch.rand(shape, device="cuda"), torch.rand(shape, device="cuda")
    t = [v.clone().requires_grad_(True) for v in (lo, pred, p, pf)]
    r = [v.clone().double().requires_grad_(True) for v in (lo, pred, p, pf)]
    final, consis = ops.joint_loss(t[0], t[1], t[2], t[3], axis, alpha, 0.1)
    consis_ref = F.mse_loss(r[2], r[3].flip(axis))
    final_ref = alpha * r[0] + (1 - alpha) * r[1] + 0.1 * consis_ref
    assert final.shape == (B, 1) and consis.shape == ()
    assert torch.allclose(consis.double(), consis_ref, rtol=1e-6, atol=1e-9)
    assert torch.allclose(final.double(), final_ref, rtol=1e-6, atol=1e-7)
    g = torch.rand(B, 1, device="cuda")
    final.backward(g)
    final_ref.backward(g.double())
    for a, b, name in zip(t, r, ("loss_out", "pred", "p", "pf")):
        assert a.grad.shape == a.shape, name
        assert torch.allclose(a.grad.double(), b.grad, rtol=2e-6, atol=1e-9), (name, float((a.grad.double() - b.grad).abs().max()))
