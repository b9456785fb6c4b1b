"""torch autograd operators over the libsprk.so C ABI (include/sprk.h).

Every operator here launches hand-written HIP kernels on torch's current stream through
ctypes; PyTorch only provides device memory, the stream and the autograd tape.  There is no
alternative implementation: CPU tensors or a missing library raise.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ConvEpilogue, ConvGeom, check  # noqa: F401


def _stream(t):
    """torch's current stream on the device that holds ``t`` (a kernel must be enqueued on a stream of the
    device its pointers live on).  One process drives one GPU here; a tensor on another device than the
    process's current one is a set-up error and raises instead of launching on the wrong card."""
    dev = t.device
    if dev.index is not None and dev.index != torch.cuda.current_device():
        raise _lib.SprkError("tensor on %s but the current device is cuda:%d — call torch.cuda.set_device(%d) "
                             "(Denoiser / DenoiserTrainer / DenoiserEvaluator do it for their own device)"
                             % (dev, torch.cuda.current_device(), dev.index))
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _need_gpu(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise _lib.SprkError("spr_pick_amd ops need float32 tensors on the GPU (got %s on %s); "
                                 "there is no CPU path" % (t.dtype, t.device))


_GRAD_DEST = None


def set_grad_destinations(flat):
    """``flat``: an object with ``dest(param) -> tensor`` (graph_step.FlatGrads) or None.  While set, the backward
    kernels write parameter gradients into the tensors it hands out (slices of one flat buffer) instead of fresh
    allocations."""
    global _GRAD_DEST
    _GRAD_DEST = flat


def _grad_like(w):
    """Where the gradient of parameter ``w`` is written: its slice of the flat gradient buffer when one is
    registered (as a fresh view object, so that autograd adopts it as ``w.grad`` without a copy), else new memory."""
    if _GRAD_DEST is not None:
        return _GRAD_DEST.dest(w)
    return torch.empty_like(w)


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


def conv_out_size(n, k, stride, dil, pad_lo, pad_hi):
    return (n + pad_lo + pad_hi - dil * (k - 1) - 1) // stride + 1


def make_geom(x, x2, w, up1, stride, dil, pad, out_hw=None, dtype=0):
    """pad = (top, bottom, left, right) zero padding of the (virtual) conv input."""
    N, C1 = x.shape[0], x.shape[1]
    Hin, Win = (x.shape[2] * 2, x.shape[3] * 2) if up1 else (x.shape[2], x.shape[3])
    C2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None and (x2.shape[0] != N or x2.shape[2] != Hin or x2.shape[3] != Win):
        raise ValueError("conv2d: skip tensor %s does not match input %s" % (tuple(x2.shape), (N, C1, Hin, Win)))
    Cout, Cin, KH, KW = w.shape
    if Cin != C1 + C2:
        raise ValueError("conv2d: weight expects %d input channels, got %d+%d" % (Cin, C1, C2))
    pt, pb, pl, pr = pad
    Hout = conv_out_size(Hin, KH, stride, dil, pt, pb)
    Wout = conv_out_size(Win, KW, stride, dil, pl, pr)
    if out_hw is not None:
        Hout, Wout = out_hw
    if Hout <= 0 or Wout <= 0:
        raise ValueError("conv2d: input %dx%d too small for kernel %dx%d dil %d" % (Hin, Win, KH, KW, dil))
    return ConvGeom(N, C1, C2, Hin, Win, 1 if up1 else 0, Cout, Hout, Wout, KH, KW, stride, dil, pt, pl, int(dtype))


def conv2d_forward(x, x2, w, g, bias=None, act=ACT_NONE, scale=None, shift=None, res=None, res_off=0, up_out=False):
    """Raw forward launch (no autograd).  Returns y [N,Cout,Hout,Wout] ([N,Cout,2Hout,2Wout] with
    ``up_out``: nearest x2 upsampling fused into the store)."""
    L = _lib.lib()
    _need_gpu(x, x2, w, bias, scale, shift, res)
    m = 2 if up_out else 1
    y = torch.empty((g.N, g.Cout, g.Hout * m, g.Wout * m), dtype=torch.float32, device=x.device)
    ep = ConvEpilogue(_p(bias), _p(scale), _p(shift), _p(res),
                      0 if res is None else res.shape[2], 0 if res is None else res.shape[3], res_off, act,
                      1 if up_out else 0)
    nb = L.sprk_conv2d_fwd_ws_bytes(ctypes.byref(g))
    ws = _ws(nb, x)
    check(L.sprk_conv2d_fwd(_p(x), _p(x2), _p(w), _p(y), ctypes.byref(g), ctypes.byref(ep), _p(ws), nb, _stream(x)),
          "sprk_conv2d_fwd")
    return y


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x2, w, bias, up1, stride, dil, pad, act, up_out, dtype):
        x = x.contiguous()
        x2 = None if x2 is None else x2.contiguous()
        w = w.contiguous()
        g = make_geom(x, x2, w, up1, stride, dil, pad, dtype=dtype)
        y = conv2d_forward(x, x2, w, g, bias=bias, act=act, up_out=up_out)
        ctx.geom = g
        ctx.act = act
        ctx.up_out = up_out
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, x2, w, y if act != ACT_NONE else None, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        x, x2, w, y, bias = ctx.saved_tensors
        g = ctx.geom
        gy = gy.contiguous()
        _need_gpu(gy)
        gx = gx2 = gw = gb = None
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        # gradient w.r.t. the pre-activation output (+ bias gradient)
        up2 = 1 if ctx.up_out else 0
        if ctx.act != ACT_NONE or need_b or up2:
            if ctx.act != ACT_NONE or up2:
                gpre = torch.empty((g.N, g.Cout, g.Hout, g.Wout), dtype=torch.float32, device=gy.device)
            else:
                gpre = gy
            if need_b:
                gb = _grad_like(bias)
            nb = L.sprk_act_bwd_ws_bytes(g.N, g.Cout, g.Hout * g.Wout)
            ws = _ws(nb, gy)
            check(L.sprk_act_bwd(_p(gy), _p(y), _p(gpre) if gpre is not gy else None, _p(gb), ctx.act,
                                 g.N, g.Cout, g.Hout, g.Wout, up2, _p(ws), nb, _stream(gy)), "sprk_act_bwd")
        else:
            gpre = gy
        if ctx.needs_input_grad[2]:
            gw = _grad_like(w)
            nb = L.sprk_conv2d_bwd_weight_ws_bytes(ctypes.byref(g))
            ws = _ws(nb, gy)
            check(L.sprk_conv2d_bwd_weight(_p(x), _p(x2), _p(gpre), _p(gw), ctypes.byref(g), _p(ws), nb, _stream(x)),
                  "sprk_conv2d_bwd_weight")
        need0 = ctx.needs_input_grad[0]
        need1 = x2 is not None and ctx.needs_input_grad[1]
        if need0 or need1:
            gd, wd = g, w
            if g.C2 and not need1:
                # the skip source needs no gradient (the raw image x0 of decode_block_1): only the
                # first C1 input channels are back-propagated
                gd = ConvGeom(g.N, g.C1, 0, g.Hin, g.Win, g.up1, g.Cout, g.Hout, g.Wout, g.KH, g.KW, g.stride, g.dil,
                              g.pad_top, g.pad_left, g.dtype)
                wd = w[:, :g.C1].contiguous()
            gin = torch.empty((gd.N, gd.C1 + gd.C2, gd.Hin, gd.Win), dtype=torch.float32, device=gy.device)
            nb = L.sprk_conv2d_bwd_data_ws_bytes(ctypes.byref(gd))
            ws = _ws(nb, gy)
            check(L.sprk_conv2d_bwd_data(_p(gpre), _p(wd), _p(gin), ctypes.byref(gd), _p(ws), nb, _stream(gpre)),
                  "sprk_conv2d_bwd_data")
            if gd.C2 == 0 and not gd.up1:
                gx = gin
            else:
                gx = torch.empty_like(x)
                gx2 = torch.empty_like(x2) if gd.C2 else None
                check(L.sprk_concat_up_bwd(_p(gin), _p(gx), _p(gx2), gd.N, gd.C1, gd.C2, gd.Hin, gd.Win, gd.up1,
                                           _stream(gin)), "sprk_concat_up_bwd")
        return gx, gx2, gw, gb, None, None, None, None, None, None, None


def conv2d(x, w, bias=None, x2=None, up1=False, stride=1, dil=1, pad=(0, 0, 0, 0), act=ACT_NONE, up_out=False,
           dtype=0):
    """y = act(conv(cat(up2(x) if up1 else x, x2), w) + bias); pad = (top, bottom, left, right).
    up_out: return nearest-x2-upsampled y (the upsampling is fused into the conv's stores).
    dtype: _lib.DT_F32 / DT_BF16 / DT_F16 — precision of the MFMA operands in forward, backward-data and
    backward-weight (a request: layers without a 16-bit kernel run in fp32; tensors are fp32 either way)."""
    _need_gpu(x, x2, w, bias)
    return _Conv2dFn.apply(x, x2, w, bias, bool(up1), int(stride), int(dil), tuple(int(p) for p in pad), int(act),
                           bool(up_out), int(dtype))


# ---- U-Net plumbing -----------------------------------------------------------------------------
class _ShiftMaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, shift):
        x = x.contiguous()
        _need_gpu(x)
        N, C, H, W = x.shape
        if H % 2 or W % 2:
            raise ValueError("shift_maxpool2: odd spatial size %dx%d" % (H, W))
        y = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        check(_lib.lib().sprk_shift_maxpool2_fwd(_p(x), _p(y), N * C, H, W, shift, _stream(x)), "sprk_shift_maxpool2_fwd")
        ctx.shift = shift
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        N, C, H, W = x.shape
        gx = torch.empty_like(x)
        check(_lib.lib().sprk_shift_maxpool2_bwd(_p(gy), _p(x), _p(gx), N * C, H, W, ctx.shift, _stream(gy)),
              "sprk_shift_maxpool2_bwd")
        return gx, None


def shift_maxpool2(x, shift=1):
    return _ShiftMaxPoolFn.apply(x, int(shift))


class _Rot4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _need_gpu(x)
        B, C, H, W = x.shape
        if H != W:
            raise ValueError("rot4_stack needs square images, got %dx%d" % (H, W))
        y = torch.empty((4 * B, C, H, W), dtype=x.dtype, device=x.device)
        check(_lib.lib().sprk_rot4_stack_fwd(_p(x), _p(y), B, C, H, _stream(x)), "sprk_rot4_stack_fwd")
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        B4, C, P, _ = gy.shape
        gx = torch.empty((B4 // 4, C, P, P), dtype=gy.dtype, device=gy.device)
        check(_lib.lib().sprk_rot4_stack_bwd(_p(gy), _p(gx), B4 // 4, C, P, _stream(gy)), "sprk_rot4_stack_bwd")
        return gx


def rot4_stack(x):
    return _Rot4Fn.apply(x)


class _UnrotFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d):
        d = d.contiguous()
        _need_gpu(d)
        B4, C, P, W = d.shape
        if P != W or B4 % 4:
            raise ValueError("unrot4_shift_concat: bad shape %s" % (tuple(d.shape),))
        f = torch.empty((B4 // 4, 4 * C, P, P), dtype=d.dtype, device=d.device)
        check(_lib.lib().sprk_unrot4_shift_concat_fwd(_p(d), _p(f), B4 // 4, C, P, _stream(d)), "sprk_unrot4_fwd")
        return f

    @staticmethod
    def backward(ctx, gf):
        gf = gf.contiguous()
        B, C4, P, _ = gf.shape
        gd = torch.empty((4 * B, C4 // 4, P, P), dtype=gf.dtype, device=gf.device)
        check(_lib.lib().sprk_unrot4_shift_concat_bwd(_p(gf), _p(gd), B, C4 // 4, P, _stream(gf)), "sprk_unrot4_bwd")
        return gd


def unrot4_shift_concat(d):
    return _UnrotFn.apply(d)


# ---- BatchNorm ------------------------------------------------------------------------------------
class _BNTrainFn(torch.autograd.Function):
    """Training-mode BatchNorm2d (+ ReLU).  ``groups`` > 1: the batch is `groups` independent passes stacked
    along N — every group gets its own batch statistics and the running averages are updated group after
    group, exactly as if the module had been called once per pass (the convolutions around it can then run
    once on the stacked batch)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, groups):
        x = x.contiguous()
        _need_gpu(x, gamma, beta)
        N, C, H, W = x.shape
        if N % groups:
            raise ValueError("batch_norm_train: batch %d is not divisible into %d groups" % (N, groups))
        Ng = N // groups
        y = torch.empty_like(x)
        mean = torch.empty((groups, C), dtype=torch.float32, device=x.device)
        invstd = torch.empty((groups, C), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        nb = L.sprk_bn_ws_bytes(Ng, C, H * W)
        ws = _ws(nb, x)
        for g in range(groups):
            sl = slice(g * Ng, (g + 1) * Ng)
            check(L.sprk_bn_train_fwd(_p(x[sl]), _p(y[sl]), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                      _p(mean[g]), _p(invstd[g]), Ng, C, H * W, momentum, eps, int(relu), _p(ws), nb,
                                      _stream(x)), "sprk_bn_train_fwd")
        ctx.relu, ctx.groups = relu, groups
        ctx.save_for_backward(x, y, gamma, mean, invstd, beta)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, mean, invstd, beta = ctx.saved_tensors
        gy = gy.contiguous()
        N, C, H, W = x.shape
        groups = ctx.groups
        Ng = N // groups
        gx = torch.empty_like(x)
        if groups == 1:     # straight into the parameters' gradient tensors
            gg, gb = _grad_like(gamma).view(1, C), _grad_like(beta).view(1, C)
        else:
            gg = torch.empty((groups, C), dtype=torch.float32, device=x.device)
            gb = torch.empty((groups, C), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        nb = L.sprk_bn_ws_bytes(Ng, C, H * W)
        ws = _ws(nb, x)
        for g in range(groups):
            sl = slice(g * Ng, (g + 1) * Ng)
            check(L.sprk_bn_train_bwd(_p(gy[sl]), _p(x[sl]), _p(y[sl]), _p(gamma), _p(mean[g]), _p(invstd[g]),
                                      _p(gx[sl]), _p(gg[g]), _p(gb[g]), Ng, C, H * W, int(ctx.relu), _p(ws), nb,
                                      _stream(gy)), "sprk_bn_train_bwd")
        if groups > 1:
            gg, gb = torch.sum(gg, 0, out=_grad_like(gamma)), torch.sum(gb, 0, out=_grad_like(beta))
        else:
            gg, gb = gg.view(C), gb.view(C)
        return gx, gg, gb, None, None, None, None, None, None


def batch_norm_train(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, relu=False, groups=1):
    return _BNTrainFn.apply(x, gamma, beta, running_mean, running_var, float(momentum), float(eps), bool(relu),
                            int(groups))


def batch_norm_eval(x, gamma, beta, running_mean, running_var, eps=1e-5, relu=False):
    """Inference-only (no autograd)."""
    x = x.contiguous()
    _need_gpu(x)
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    check(_lib.lib().sprk_bn_eval_fwd(_p(x), _p(y), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                      N, C, H * W, float(eps), int(relu), _stream(x)), "sprk_bn_eval_fwd")
    return y


# ---- per-pixel pipeline maths -------------------------------------------------------------------
class _ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out_stats, eps):
        out_stats = out_stats.contiguous()
        eps = eps.contiguous()
        _need_gpu(out_stats, eps)
        B, C, H, W = out_stats.shape
        if C != 2:
            raise ValueError("reparameterize expects out_stats with 2 channels")
        z = torch.empty((B, 1, H, W), dtype=out_stats.dtype, device=out_stats.device)
        check(_lib.lib().sprk_reparam_fwd(_p(out_stats), _p(eps), _p(z), B, H * W, _stream(out_stats)), "sprk_reparam_fwd")
        ctx.save_for_backward(out_stats, eps)
        return z

    @staticmethod
    def backward(ctx, gz):
        out_stats, eps = ctx.saved_tensors
        gz = gz.contiguous()
        B, _, H, W = out_stats.shape
        go = torch.empty_like(out_stats)
        check(_lib.lib().sprk_reparam_bwd(_p(gz), _p(out_stats), _p(eps), _p(go), B, H * W, _stream(gz)), "sprk_reparam_bwd")
        return go, None


def reparameterize(out_stats, eps):
    return _ReparamFn.apply(out_stats, eps)


class _SigmoidClampFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _need_gpu(x)
        p = torch.empty_like(x)
        check(_lib.lib().sprk_sigmoid_clamp_fwd(_p(x), _p(p), x.numel(), _stream(x)), "sprk_sigmoid_clamp_fwd")
        ctx.save_for_backward(x)
        return p

    @staticmethod
    def backward(ctx, gp):
        (x,) = ctx.saved_tensors
        gp = gp.contiguous()
        gx = torch.empty_like(x)
        check(_lib.lib().sprk_sigmoid_clamp_bwd(_p(gp), _p(x), _p(gx), x.numel(), _stream(gp)), "sprk_sigmoid_clamp_bwd")
        return gx


def sigmoid_clamp(x):
    return _SigmoidClampFn.apply(x)


class _SsdnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_stats, noise_std):
        x = x.contiguous()
        out_stats = out_stats.contiguous()
        ns = noise_std.reshape(-1).contiguous()
        _need_gpu(x, out_stats, ns)
        B, _, H, W = out_stats.shape
        L = _lib.lib()
        loss = torch.empty((B, 1), dtype=torch.float32, device=x.device)
        pme = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
        mstd = torch.empty((1, B, H, W), dtype=torch.float32, device=x.device)
        nb = L.sprk_ssdn_ws_bytes(B, H * W)
        ws = _ws(nb, x)
        check(L.sprk_ssdn_fwd(_p(x), _p(out_stats), _p(ns), _p(loss), _p(pme), _p(mstd), B, H * W, _p(ws), nb, _stream(x)),
              "sprk_ssdn_fwd")
        ctx.save_for_backward(x, out_stats, ns)
        ctx.ns_shape = noise_std.shape
        ctx.mark_non_differentiable(pme, mstd)
        return loss, pme, mstd

    @staticmethod
    def backward(ctx, gloss, _gpme, _gmstd):
        x, out_stats, ns = ctx.saved_tensors
        B, _, H, W = out_stats.shape
        L = _lib.lib()
        gl = gloss.reshape(-1).contiguous()
        go = torch.empty_like(out_stats)
        gns = torch.empty(B, dtype=torch.float32, device=x.device)
        nb = L.sprk_ssdn_ws_bytes(B, H * W)
        ws = _ws(nb, x)
        check(L.sprk_ssdn_bwd(_p(gl), _p(x), _p(out_stats), _p(ns), _p(go), _p(gns), B, H * W, _p(ws), nb, _stream(gl)),
              "sprk_ssdn_bwd")
        return None, go, gns.reshape(ctx.ns_shape)


def ssdn_nll_pme(x, out_stats, noise_std):
    """(loss [B,1] = per-image mean NLL, pme [B,1,H,W], model_std [1,B,H,W])."""
    return _SsdnFn.apply(x, out_stats, noise_std)
