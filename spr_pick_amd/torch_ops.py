"""``torch.ops.sprk.*``: the libsprk.so entry points registered as PyTorch custom operators (torch.library).

One operator per C-ABI function of include/sprk.h, schema-typed, with a CUDA implementation (the ctypes call on
torch's current stream of the tensors' device) and a fake / meta implementation (output shapes only), so the
dispatcher, ``torch.library.opcheck``-style tooling, FakeTensor tracing and CUDA-graph capture all see them as ordinary
operators.  Autograd is NOT registered here: ops.py's ``autograd.Function`` classes call these operators in their
forward and backward (the backward kernels are operators too), which keeps a single place for the saved-tensor policy.
There is no CPU implementation — dispatching a CPU tensor raises — and a missing libsprk.so raises at first use.

Geometry travels as the 16 integers of ``sprk_conv_geom`` (``ConvGeom`` field order: N, C1, C2, Hin, Win, up1, Cout,
Hout, Wout, KH, KW, stride, dil, pad_top, pad_left, dtype).
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvEpilogue, ConvGeom, check

_LIB = torch.library.Library("sprk", "DEF")
_NAMES = []


def _stream(t):
    dev = t.device
    if dev.index is not None and dev.index != torch.cuda.current_device():
        raise _lib.SprkError("tensor on %s but the current device is cuda:%d — call torch.cuda.set_device(%d) "
                             "(Denoiser / DenoiserTrainer / DenoiserEvaluator do it for their own device)"
                             % (dev, torch.cuda.current_device(), dev.index))
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


def _f32(like, shape):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# storage types of activation tensors: SPRK_DT_* codes <-> torch dtypes (include/sprk.h: SPRK_IO2 / SPRK_IO3)
TORCH_OF = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}
CODE_OF = {v: k for k, v in TORCH_OF.items()}


def code(t):
    c = CODE_OF.get(t.dtype)
    if c is None:
        raise _lib.SprkError("activation tensors are float32, bfloat16 or float16 (got %s)" % t.dtype)
    return c


def _alloc(like, shape, c):
    return torch.empty(shape, dtype=TORCH_OF[c], device=like.device)


def _out_code(geom_dtype):
    """storage type of a convolution call's output tensor: the operand type with SPRK_DT_Y16, else fp32"""
    return (geom_dtype & 0xff) if (geom_dtype & _lib.DT_Y16) else 0


def _register(name, schema, impl, fake):
    _LIB.define(name + schema)
    _LIB.impl(name, impl, "CUDA")
    torch.library.register_fake("sprk::" + name, fake, lib=_LIB)
    _NAMES.append(name)


def geom_list(g):
    return [getattr(g, f) for f, _ in ConvGeom._fields_]


# ---- convolution -----------------------------------------------------------------------------------------------------
def _epilogue(bias, scale, shift, res, res_off, act, up_out):
    return ConvEpilogue(_p(bias), _p(scale), _p(shift), _p(res), 0 if res is None else res.shape[2],
                        0 if res is None else res.shape[3], res_off, act, 1 if up_out else 0)


def _conv2d_fwd(x, x2, w, bias, scale, shift, res, geom, res_off, act, up_out, ws):
    """ws: a caller-owned workspace (prepared weights, ops.WeightPrep) or None = a fresh one."""
    L = _lib.lib()
    g = ConvGeom(*geom)
    m = 2 if up_out else 1
    y = _alloc(x, (g.N, g.Cout, g.Hout * m, g.Wout * m), _out_code(g.dtype))
    ep = _epilogue(bias, scale, shift, res, res_off, act, up_out)
    if ws is None:
        ws = _ws(L.sprk_conv2d_fwd_ws_bytes(ctypes.byref(g)), x)
    check(L.sprk_conv2d_fwd(_p(x), _p(x2), _p(w), _p(y), ctypes.byref(g), ctypes.byref(ep), _p(ws), ws.numel(), _stream(x)),
          "sprk_conv2d_fwd")
    return y


def _conv2d_fwd_fake(x, x2, w, bias, scale, shift, res, geom, res_off, act, up_out, ws):
    m = 2 if up_out else 1
    return x.new_empty((geom[0], geom[6], geom[7] * m, geom[8] * m), dtype=TORCH_OF[_out_code(geom[15])])


_register("conv2d_fwd", "(Tensor x, Tensor? x2, Tensor w, Tensor? bias, Tensor? scale, Tensor? shift, Tensor? res, "
                        "int[] geom, int res_off, int act, int up_out, Tensor? ws) -> Tensor", _conv2d_fwd, _conv2d_fwd_fake)


def _conv2d_bwd_data(gy, w, geom, mask_y, mask_act, ws):
    """mask_y / mask_act: fuse the activation backward of the layer that produced this convolution's input (the saved
    input itself is the mask): the result is that layer's pre-activation gradient (sprk_conv2d_bwd_data_masked).
    ws: a caller-owned workspace (prepared weights) or None."""
    L = _lib.lib()
    g = ConvGeom(*geom)
    gin = _alloc(gy, (g.N, g.C1 + g.C2, g.Hin, g.Win), _out_code(g.dtype))
    if ws is None:
        ws = _ws(L.sprk_conv2d_bwd_data_ws_bytes(ctypes.byref(g)), gy)
    if mask_y is not None and mask_y.dtype != gin.dtype:
        raise _lib.SprkError("conv2d_bwd_data: the mask (%s) must have the storage type of the input gradient (%s)" % (
            mask_y.dtype, gin.dtype))
    if mask_y is not None and tuple(mask_y.shape) != tuple(gin.shape):
        raise _lib.SprkError("conv2d_bwd_data: mask %s does not match the input gradient %s" % (tuple(mask_y.shape), tuple(gin.shape)))
    check(L.sprk_conv2d_bwd_data_masked(_p(gy), _p(w), _p(gin), ctypes.byref(g), _p(mask_y), int(mask_act) if mask_y is not None else 0,
                                        _p(ws), ws.numel(), _stream(gy)), "sprk_conv2d_bwd_data_masked")
    return gin


_register("conv2d_bwd_data", "(Tensor gy, Tensor w, int[] geom, Tensor? mask_y, int mask_act, Tensor? ws) -> Tensor",
          _conv2d_bwd_data,
          lambda gy, w, geom, mask_y, mask_act, ws: gy.new_empty((geom[0], geom[1] + geom[2], geom[3], geom[4]),
                                                                  dtype=TORCH_OF[_out_code(geom[15])]))


def _head1x1_fwd(f, w1, b1, w2, b2, w3, b3):
    """Fused per-pixel head (inference): [B,K0,H,W] -> [B,N3,H,W] (sprk_head1x1_fwd)."""
    L = _lib.lib()
    B, K0, H, W = f.shape
    N1, N3 = w1.shape[0], w3.shape[0]
    out = _f32(f, (B, N3, H, W))
    nb = L.sprk_head1x1_fwd_ws_bytes(K0, N1)
    if nb == 0:
        raise _lib.SprkError("head1x1_fwd: no fused kernel for %d -> %d" % (K0, N1))
    ws = _ws(nb, f)
    check(L.sprk_head1x1_fwd(_p(f), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(out), B, K0, N1, N3, H * W, _p(ws), nb,
                             _stream(f)), "sprk_head1x1_fwd")
    return out


_register("head1x1_fwd", "(Tensor f, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor w3, Tensor b3) -> Tensor", _head1x1_fwd,
          lambda f, w1, b1, w2, b2, w3, b3: f.new_empty((f.shape[0], w3.shape[0], f.shape[2], f.shape[3])))


def _head1x1_unrot_fwd(d, w1, b1, w2, b2, w3, b3):
    """Blind-spot head on the rotated stack: d [4B,96,P,P] -> [B,N3,P,P] (sprk_head1x1_unrot_fwd)."""
    L = _lib.lib()
    B4, C, P, P2 = d.shape
    if B4 % 4 or P != P2:
        raise _lib.SprkError("head1x1_unrot_fwd: bad shape %s" % (tuple(d.shape),))
    N3 = w3.shape[0]
    out = _f32(d, (B4 // 4, N3, P, P))
    nb = L.sprk_head1x1_fwd_ws_bytes(384, 384)
    ws = _ws(nb, d)
    check(L.sprk_head1x1_unrot_fwd(_p(d), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(out), B4 // 4, C, P, N3, _p(ws), nb,
                                   _stream(d)), "sprk_head1x1_unrot_fwd")
    return out


_register("head1x1_unrot_fwd", "(Tensor d, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor w3, Tensor b3) -> Tensor",
          _head1x1_unrot_fwd,
          lambda d, w1, b1, w2, b2, w3, b3: d.new_empty((d.shape[0] // 4, w3.shape[0], d.shape[2], d.shape[3])))


# Pending second-stage sums (sprk_reduce_items): with defer=True the backward-weight / bias-gradient operators run only
# their main kernel, and the ~80 small sums of a training step are finished together by ``reduce_pending``.  An entry
# keeps its partial buffer alive until then; the destination must be kept alive by the caller (ops defers only
# gradients that live in graph_step.FlatGrads' buffer).
_PENDING = {}


def _pend(dev, item, *keep):
    if item.kind != 0:
        _PENDING.setdefault(dev.index, []).append((item, keep))


def pending_count(dev):
    return len(_PENDING.get(dev.index, ()))


def _conv2d_bwd_weight(x, x2, gy, geom, gw, defer):
    """Out variant: writes into ``gw`` (a fresh tensor or the parameter's slice of the flat gradient buffer).
    defer: leave the sum over the workgroups' partial results to ``reduce_pending`` (gw is undefined until then)."""
    L = _lib.lib()
    g = ConvGeom(*geom)
    nb = L.sprk_conv2d_bwd_weight_ws_bytes(ctypes.byref(g))
    ws = _ws(nb, gy)
    if defer:
        item = _lib.ReduceItem()
        check(L.sprk_conv2d_bwd_weight_partial(_p(x), _p(x2), _p(gy), _p(gw), ctypes.byref(g), _p(ws), nb, ctypes.byref(item),
                                               _stream(x)), "sprk_conv2d_bwd_weight_partial")
        _pend(gy.device, item, ws)     # not gw: autograd adopts a gradient tensor only while nobody else holds it
        return
    check(L.sprk_conv2d_bwd_weight(_p(x), _p(x2), _p(gy), _p(gw), ctypes.byref(g), _p(ws), nb, _stream(x)),
          "sprk_conv2d_bwd_weight")


_register("conv2d_bwd_weight", "(Tensor x, Tensor? x2, Tensor gy, int[] geom, Tensor(a!) gw, bool defer) -> ()",
          _conv2d_bwd_weight, lambda x, x2, gy, geom, gw, defer: None)


def _reduce_pending(like):
    """Finish every pending sum of ``like``'s device in one launch per 48 items."""
    items = _PENDING.pop(like.device.index, [])
    if not items:
        return
    arr = (_lib.ReduceItem * len(items))(*[it for it, _ in items])
    check(_lib.lib().sprk_reduce_items(arr, len(items), _stream(like)), "sprk_reduce_items")


_register("reduce_pending", "(Tensor like) -> ()", _reduce_pending, lambda like: None)


def drop_pending(dev):
    """Forget pending sums without running them (error paths)."""
    _PENDING.pop(dev.index, None)


def _act_bwd(gy, y, act, geom4, up2, want_gpre, gbias, defer, out_code):
    """gpre = gy * act'(y) (2x2-summed first when up2); bias gradient into ``gbias`` when given (defer: its final
    sum is left to ``reduce_pending``).  Returns gpre (or gy itself when no new tensor is needed).  out_code: storage
    type of gpre (SPRK_DT_*; gy, y and gpre may each be fp32 or the 16-bit type)."""
    L = _lib.lib()
    N, C, H, W = geom4
    gpre = _alloc(gy, (N, C, H, W), out_code) if want_gpre else None
    io = code(gy) | ((code(y) if y is not None else 0) << 4) | ((out_code if want_gpre else code(gy)) << 8)
    # gy may be a channel slice of a wider tensor (dense planes, a larger stride between images): read in place
    gstride = 0
    if not gy.is_contiguous():
        if want_gpre and gy.dim() == 4 and gy[0].is_contiguous() and gy.stride(0) >= gy[0].numel():
            gstride = gy.stride(0)
        else:
            gy = gy.contiguous()
    nb = L.sprk_act_bwd_ws_bytes(N, C, H * W)
    ws = _ws(nb, gy)
    if defer and gbias is not None:
        item = _lib.ReduceItem()
        check(L.sprk_act_bwd_partial(_p(gy), _p(y), _p(gpre), _p(gbias), act, N, C, H, W, up2, gstride, io, _p(ws), nb,
                                     ctypes.byref(item), _stream(gy)), "sprk_act_bwd_partial")
        _pend(gy.device, item, ws)
    else:
        check(L.sprk_act_bwd(_p(gy), _p(y), _p(gpre), _p(gbias), act, N, C, H, W, up2, gstride, io, _p(ws), nb, _stream(gy)),
              "sprk_act_bwd")
    return gpre if want_gpre else gy.new_empty(0)


_register("act_bwd", "(Tensor gy, Tensor? y, int act, int[] nchw, int up2, bool want_gpre, Tensor(a!)? gbias, bool defer, "
                     "int out_code) -> Tensor", _act_bwd,
          lambda gy, y, act, nchw, up2, want_gpre, gbias, defer, out_code: gy.new_empty(
              tuple(nchw) if want_gpre else (0,), dtype=TORCH_OF[out_code] if want_gpre else gy.dtype))


def _concat_up_bwd(gin, C1, C2, up1, x_shape, x2_shape):
    L = _lib.lib()
    N, _, H, W = gin.shape
    gx = _f32(gin, tuple(x_shape))
    gx2 = _f32(gin, tuple(x2_shape)) if C2 else gin.new_empty(0)
    check(L.sprk_concat_up_bwd(_p(gin), _p(gx), _p(gx2) if C2 else None, N, C1, C2, H, W, up1, _stream(gin)),
          "sprk_concat_up_bwd")
    return gx, gx2


_register("concat_up_bwd", "(Tensor gin, int C1, int C2, int up1, int[] x_shape, int[] x2_shape) -> (Tensor, Tensor)",
          _concat_up_bwd, lambda gin, C1, C2, up1, xs, x2s: (gin.new_empty(tuple(xs)), gin.new_empty(tuple(x2s) if C2 else (0,))))


# ---- U-Net plumbing ----------------------------------------------------------------------------------------------------
def _simple(name, schema, cfn, out_shape, args):
    """Register an operator whose C function takes (inputs..., output, ints..., stream)."""
    def impl(*a):
        L = _lib.lib()
        tensors = [t for t in a if torch.is_tensor(t)]
        y = _f32(tensors[0], out_shape(*a))
        check(getattr(L, cfn)(*args(a, y), _stream(tensors[0])), cfn)
        return y
    _register(name, schema, impl, lambda *a: [t for t in a if torch.is_tensor(t)][0].new_empty(out_shape(*a)))


def _typed(name, schema, cfn, out_shape, out_dtype, args):
    """Like _simple for the operators whose tensors may be fp32 or 16-bit: out_dtype(*a) -> torch dtype of the output,
    args(a, y) -> C arguments incl. the io word."""
    def impl(*a):
        L = _lib.lib()
        tensors = [t for t in a if torch.is_tensor(t)]
        y = torch.empty(out_shape(*a), dtype=out_dtype(*a), device=tensors[0].device)
        check(getattr(L, cfn)(*args(a, y), _stream(tensors[0])), cfn)
        return y
    _register(name, schema, impl, lambda *a: [t for t in a if torch.is_tensor(t)][0].new_empty(out_shape(*a), dtype=out_dtype(*a)))


_typed("shift_maxpool2_fwd", "(Tensor x, int shift) -> Tensor", "sprk_shift_maxpool2_fwd",
       lambda x, shift: (x.shape[0], x.shape[1], x.shape[2] // 2, x.shape[3] // 2), lambda x, shift: x.dtype,
       lambda a, y: (_p(a[0]), _p(y), a[0].shape[0] * a[0].shape[1], a[0].shape[2], a[0].shape[3], a[1],
                     code(a[0]) | (code(y) << 4)))
_typed("shift_maxpool2_bwd", "(Tensor gy, Tensor x, int shift, int act) -> Tensor", "sprk_shift_maxpool2_bwd",
       lambda gy, x, shift, act: tuple(x.shape), lambda gy, x, shift, act: x.dtype,
       lambda a, y: (_p(a[0]), _p(a[1]), _p(y), a[1].shape[0] * a[1].shape[1], a[1].shape[2], a[1].shape[3], a[2], a[3],
                     code(a[0]) | (code(a[1]) << 4) | (code(y) << 8)))
_simple("rot4_stack_fwd", "(Tensor x) -> Tensor", "sprk_rot4_stack_fwd",
        lambda x: (4 * x.shape[0], x.shape[1], x.shape[2], x.shape[3]),
        lambda a, y: (_p(a[0]), _p(y), a[0].shape[0], a[0].shape[1], a[0].shape[2]))
_simple("rot4_stack_bwd", "(Tensor gy) -> Tensor", "sprk_rot4_stack_bwd",
        lambda gy: (gy.shape[0] // 4, gy.shape[1], gy.shape[2], gy.shape[3]),
        lambda a, y: (_p(a[0]), _p(y), a[0].shape[0] // 4, a[0].shape[1], a[0].shape[2]))
_typed("unrot4_shift_concat_fwd", "(Tensor d) -> Tensor", "sprk_unrot4_shift_concat_fwd",
       lambda d: (d.shape[0] // 4, 4 * d.shape[1], d.shape[2], d.shape[3]), lambda d: d.dtype,
       lambda a, y: (_p(a[0]), _p(y), a[0].shape[0] // 4, a[0].shape[1], a[0].shape[2], code(a[0]) | (code(y) << 4)))
_typed("unrot4_shift_concat_bwd", "(Tensor gf) -> Tensor", "sprk_unrot4_shift_concat_bwd",
       lambda gf: (4 * gf.shape[0], gf.shape[1] // 4, gf.shape[2], gf.shape[3]), lambda gf: gf.dtype,
       lambda a, y: (_p(a[0]), _p(y), a[0].shape[0], a[0].shape[1] // 4, a[0].shape[2], code(a[0]) | (code(y) << 4)))

# ---- per-pixel pipeline maths ----------------------------------------------------------------------------------------------
_simple("reparam_fwd", "(Tensor out_stats, Tensor eps) -> Tensor", "sprk_reparam_fwd",
        lambda o, e: (o.shape[0], 1, o.shape[2], o.shape[3]),
        lambda a, y: (_p(a[0]), _p(a[1]), _p(y), a[0].shape[0], a[0].shape[2] * a[0].shape[3]))
_simple("reparam_bwd", "(Tensor gz, Tensor out_stats, Tensor eps) -> Tensor", "sprk_reparam_bwd",
        lambda gz, o, e: tuple(o.shape),
        lambda a, y: (_p(a[0]), _p(a[1]), _p(a[2]), _p(y), a[1].shape[0], a[1].shape[2] * a[1].shape[3]))
_simple("sigmoid_clamp_fwd", "(Tensor x) -> Tensor", "sprk_sigmoid_clamp_fwd", lambda x: tuple(x.shape),
        lambda a, y: (_p(a[0]), _p(y), a[0].numel()))
_simple("sigmoid_clamp_bwd", "(Tensor gp, Tensor x) -> Tensor", "sprk_sigmoid_clamp_bwd", lambda gp, x: tuple(x.shape),
        lambda a, y: (_p(a[0]), _p(a[1]), _p(y), a[1].numel()))


# ---- fused pieces of the training step's tail (include/sprk.h, ABI 410) ---------------------------------------------------
def _crop_shape(x, off, stride, out_h, out_w):
    return (x.shape[0], x.shape[1], out_h, out_w)


def _crop_add(y, x, off, stride, out_h, out_w):
    out = _f32(x, _crop_shape(x, off, stride, out_h, out_w))
    check(_lib.lib().sprk_crop_add_fwd(_p(y), _p(x), _p(out), x.shape[0] * x.shape[1], out_h, out_w, x.shape[2], x.shape[3],
                                       off, stride, _stream(x)), "sprk_crop_add_fwd")
    return out


_register("crop_add_fwd", "(Tensor? y, Tensor x, int off, int stride, int out_h, int out_w) -> Tensor", _crop_add,
          lambda y, x, off, stride, out_h, out_w: x.new_empty(_crop_shape(x, off, stride, out_h, out_w)))
_simple("crop_embed_bwd", "(Tensor g, int off, int stride, int hx, int wx) -> Tensor", "sprk_crop_embed_bwd",
        lambda g, off, stride, hx, wx: (g.shape[0], g.shape[1], hx, wx),
        lambda a, y: (_p(a[0]), _p(y), a[0].shape[0] * a[0].shape[1], a[0].shape[2], a[0].shape[3], a[3], a[4], a[1], a[2]))


def _noise_std_fwd(est):
    B, HW = est.shape[0], est[0].numel()
    out, z = _f32(est, (B, 1, 1, 1)), _f32(est, (B,))
    check(_lib.lib().sprk_noise_std_fwd(_p(est), _p(out), _p(z), B, HW, _stream(est)), "sprk_noise_std_fwd")
    return out, z


_register("noise_std_fwd", "(Tensor est) -> (Tensor, Tensor)", _noise_std_fwd,
          lambda est: (est.new_empty((est.shape[0], 1, 1, 1)), est.new_empty((est.shape[0],))))
_simple("noise_std_bwd", "(Tensor g, Tensor z, int c, int h, int w) -> Tensor", "sprk_noise_std_bwd",
        lambda g, z, c, h, w: (z.shape[0], c, h, w),
        lambda a, y: (_p(a[0]), _p(a[1]), _p(y), a[1].shape[0], a[2] * a[3] * a[4]))


def _joint_loss_fwd(loss_out, pred, p, pf, axis, alpha, wc):
    B, H, W = p.shape[0], p.shape[2], p.shape[3]
    final, consis = _f32(p, (B, 1)), _f32(p, (1,))
    check(_lib.lib().sprk_joint_loss_fwd(_p(loss_out), _p(pred), _p(p), _p(pf), _p(final), _p(consis), B, H, W, axis,
                                         ctypes.c_float(alpha), ctypes.c_float(wc), _stream(p)), "sprk_joint_loss_fwd")
    return final, consis


def _joint_loss_bwd(g, p, pf, axis, alpha, wc):
    B, H, W = p.shape[0], p.shape[2], p.shape[3]
    gl, gpred, gp, gpf = _f32(p, (B, 1)), _f32(p, (1,)), _f32(p, tuple(p.shape)), _f32(p, tuple(p.shape))
    check(_lib.lib().sprk_joint_loss_bwd(_p(g), _p(p), _p(pf), _p(gl), _p(gpred), _p(gp), _p(gpf), B, H, W, axis,
                                         ctypes.c_float(alpha), ctypes.c_float(wc), _stream(p)), "sprk_joint_loss_bwd")
    return gl, gpred, gp, gpf


_register("joint_loss_fwd", "(Tensor loss_out, Tensor pred, Tensor p, Tensor pf, int axis, float alpha, float wc) -> (Tensor, Tensor)",
          _joint_loss_fwd, lambda lo, pred, p, pf, axis, alpha, wc: (p.new_empty((p.shape[0], 1)), p.new_empty((1,))))
_register("joint_loss_bwd", "(Tensor g, Tensor p, Tensor pf, int axis, float alpha, float wc) -> (Tensor, Tensor, Tensor, Tensor)",
          _joint_loss_bwd, lambda g, p, pf, axis, alpha, wc: (p.new_empty((p.shape[0], 1)), p.new_empty((1,)),
                                                             p.new_empty(p.shape), p.new_empty(p.shape)))


def _pu_loss(p, y, log_binom, slack):
    """-> (loss [1], d loss / d p [B]) in one launch."""
    B = p.numel()
    if log_binom.shape != (B + 1, B + 1) or y.numel() != B:
        raise ValueError("pu_loss: table %s / labels %d do not match %d scores" % (tuple(log_binom.shape), y.numel(), B))
    loss, gp = _f32(p, (1,)), _f32(p, (B,))
    check(_lib.lib().sprk_pu_loss(_p(p), _p(y), _p(log_binom), B, ctypes.c_float(slack), _p(loss), _p(gp), _stream(p)),
          "sprk_pu_loss")
    return loss, gp


_register("pu_loss", "(Tensor p, Tensor y, Tensor log_binom, float slack) -> (Tensor, Tensor)", _pu_loss,
          lambda p, y, t, slack: (p.new_empty((1,)), p.new_empty((p.numel(),))))


def _ssdn_fwd(x, out_stats, noise_std, style):
    """-> (loss [B,1], pme [B,1,H,W], model_std [1,B,H,W], noise_std_map [B,H,W] (poisson) or empty (gaussian))"""
    L = _lib.lib()
    B, _, H, W = out_stats.shape
    loss, pme, mstd = _f32(x, (B, 1)), _f32(x, (B, 1, H, W)), _f32(x, (1, B, H, W))
    nsmap = _f32(x, (B, H, W)) if style == 1 else _f32(x, (0,))
    nb = L.sprk_ssdn_ws_bytes(B, H * W)
    ws = _ws(nb, x)
    check(L.sprk_ssdn_fwd(_p(x), _p(out_stats), _p(noise_std), _p(loss), _p(pme), _p(mstd), _p(nsmap) if style == 1 else None,
                          int(style), B, H * W, _p(ws), nb, _stream(x)), "sprk_ssdn_fwd")
    return loss, pme, mstd, nsmap


def _ssdn_fwd_fake(x, out_stats, noise_std, style):
    B, _, H, W = out_stats.shape
    return (x.new_empty((B, 1)), x.new_empty((B, 1, H, W)), x.new_empty((1, B, H, W)),
            x.new_empty((B, H, W) if style == 1 else (0,)))


_register("ssdn_fwd", "(Tensor x, Tensor out_stats, Tensor noise_std, int style) -> (Tensor, Tensor, Tensor, Tensor)", _ssdn_fwd,
          _ssdn_fwd_fake)


def _ssdn_bwd(gloss, x, out_stats, noise_std, style):
    L = _lib.lib()
    B, _, H, W = out_stats.shape
    go, gns = torch.empty_like(out_stats), _f32(x, (B,))
    nb = L.sprk_ssdn_ws_bytes(B, H * W)
    ws = _ws(nb, x)
    check(L.sprk_ssdn_bwd(_p(gloss), _p(x), _p(out_stats), _p(noise_std), _p(go), _p(gns), int(style), B, H * W, _p(ws), nb,
                          _stream(x)), "sprk_ssdn_bwd")
    return go, gns


_register("ssdn_bwd", "(Tensor gloss, Tensor x, Tensor out_stats, Tensor noise_std, int style) -> (Tensor, Tensor)", _ssdn_bwd,
          lambda gl, x, o, ns, style: (torch.empty_like(o), x.new_empty((o.shape[0],))))


# ---- BatchNorm -----------------------------------------------------------------------------------------------------------
def _bn_eval_fwd(x, gamma, beta, running_mean, running_var, eps, relu):
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    check(_lib.lib().sprk_bn_eval_fwd(_p(x), _p(y), _p(gamma), _p(beta), _p(running_mean), _p(running_var), N, C, H * W,
                                      float(eps), int(relu), _stream(x)), "sprk_bn_eval_fwd")
    return y


_register("bn_eval_fwd", "(Tensor x, Tensor gamma, Tensor beta, Tensor running_mean, Tensor running_var, float eps, "
                         "bool relu) -> Tensor", _bn_eval_fwd, lambda x, *a: torch.empty_like(x))


def _bn_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu, groups):
    """`groups` passes stacked along N, each with its own batch statistics: returns (y, save_mean [groups, C],
    save_invstd [groups, C]); updates the running statistics in place, group after group."""
    L = _lib.lib()
    N, C, H, W = x.shape
    y, mean, invstd = torch.empty_like(x), _f32(x, (groups, C)), _f32(x, (groups, C))
    nb = L.sprk_bn_ws_bytes(N, C, H * W, groups)
    ws = _ws(nb, x)
    check(L.sprk_bn_train_fwd(_p(x), _p(y), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(mean), _p(invstd),
                              N, C, H * W, groups, momentum, eps, int(relu), _p(ws), nb, _stream(x)), "sprk_bn_train_fwd")
    return y, mean, invstd


_register("bn_train_fwd", "(Tensor x, Tensor gamma, Tensor beta, Tensor(a!) running_mean, Tensor(b!) running_var, "
                          "float momentum, float eps, bool relu, int groups) -> (Tensor, Tensor, Tensor)", _bn_train_fwd,
          lambda x, g, b, rm, rv, mo, eps, relu, groups: (torch.empty_like(x), g.new_empty((groups,) + tuple(g.shape)),
                                                         g.new_empty((groups,) + tuple(g.shape))))


def _bn_train_bwd(gy, x, y, gamma, mean, invstd, relu, groups, ggamma, gbeta):
    L = _lib.lib()
    N, C, H, W = x.shape
    gx = torch.empty_like(x)
    nb = L.sprk_bn_ws_bytes(N, C, H * W, groups)
    ws = _ws(nb, x)
    check(L.sprk_bn_train_bwd(_p(gy), _p(x), _p(y), _p(gamma), _p(mean), _p(invstd), _p(gx), _p(ggamma), _p(gbeta),
                              N, C, H * W, groups, int(relu), _p(ws), nb, _stream(x)), "sprk_bn_train_bwd")
    return gx


_register("bn_train_bwd", "(Tensor gy, Tensor x, Tensor y, Tensor gamma, Tensor mean, Tensor invstd, bool relu, "
                          "int groups, Tensor(a!) ggamma, Tensor(b!) gbeta) -> Tensor", _bn_train_bwd,
          lambda gy, x, *a: torch.empty_like(x))


# ---- NMS (one relaxation call; algorithms.nms_device drives the fixed point) -------------------------------------------------
def _nms2d(score, r, threshold, out_s, out_xy, cnt, rounds, resume, ws):
    H, W = score.shape
    check(_lib.lib().sprk_nms2d(_p(score), H, W, int(r), ctypes.c_float(threshold), _p(out_s), _p(out_xy), _p(cnt),
                                out_s.shape[0], rounds, resume, _p(ws), ws.numel(), _stream(score)), "sprk_nms2d")


_register("nms2d", "(Tensor score, int r, float threshold, Tensor(a!) out_s, Tensor(b!) out_xy, Tensor(c!) cnt, "
                   "int rounds, int resume, Tensor(d!) ws) -> ()", _nms2d, lambda *a: None)


def registered():
    """Names of the operators under torch.ops.sprk (tests)."""
    return tuple(_NAMES)
