"""Differentiable operators over the libsprk.so C ABI (include/sprk.h).

Two layers: torch_ops.py registers every C entry point as a PyTorch custom operator (``torch.ops.sprk.*``:
schema, CUDA implementation = the ctypes call on torch's current stream, fake implementation for shape
propagation); this module adds the autograd formulas (``autograd.Function`` classes whose forward AND backward are
``torch.ops.sprk`` calls) and the functional API the networks use.  PyTorch provides device memory, the stream, the
dispatcher and the autograd tape; there is no alternative implementation: CPU tensors or a missing library raise.
"""
import ctypes

import torch

from . import _lib
from . import torch_ops  # noqa: F401  (registers torch.ops.sprk.*)
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ConvEpilogue, ConvGeom, check  # noqa: F401
from .torch_ops import geom_list

_S = torch.ops.sprk


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


def _need_act(*ts):
    """Activation tensors: float32, or — the 16-bit storage modes — bfloat16 / float16, on the GPU."""
    for t in ts:
        if t is not None and (not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16, torch.float16)):
            raise _lib.SprkError("spr_pick_amd ops need float32 / bfloat16 / float16 activation tensors on the GPU (got %s on "
                                 "%s); there is no CPU path" % (t.dtype, t.device))


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


def _grad_dest(w):
    """-> (tensor, defer).  defer: the operator may leave the second stage of its sum pending (torch_ops.reduce_pending
    finishes all of them in one launch when the flat-gradient context closes).  Only the FIRST gradient of a parameter
    in a step is deferred: a second consumer makes autograd add the two at once, so what is pending is finished first."""
    if _GRAD_DEST is None:
        return torch.empty_like(w), False
    first = not _GRAD_DEST.handed_out(w)
    t = _GRAD_DEST.dest(w)
    if not first:
        flush_reductions(w)
    return t, first and _GRAD_DEST.defer


def flush_reductions(like):
    _S.reduce_pending(like)


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


class WeightPrep:
    """Prepared weights (include/sprk.h, SPRK_DT_WPREP): the per-call weight transforms of the convolutions hoisted
    into one launch per optimiser step.

    While the context is open, the first forward / backward-data call of a (weight tensor, geometry, epilogue shape)
    runs normally in a workspace that is KEPT, and the transform it ran is recorded as an item; from then on
    ``begin_step()`` re-runs all recorded transforms in one launch per 40 (the weights changed: call it once per step,
    before the first convolution) and the calls pass SPRK_DT_WPREP, i.e. skip their own transform launch.
    Same device code either way: results are bit-identical.  Used by graph_step.GraphedTrainStep (the launch is the
    first node of the captured graphs)."""

    def __init__(self):
        self.clear()
        self.launches = 0

    def clear(self):
        """Forget every kept workspace (the model or the stepper is being rebuilt)."""
        self.entries = {}        # key -> (workspace tensor, the weight tensor it was made from, kind of the transform)
        self.items = []          # _lib.WprepItem of every entry with a transform
        self._table = None

    def __enter__(self):
        global _WPREP
        self._outer, _WPREP = _WPREP, self
        return self

    def __exit__(self, *exc):
        global _WPREP
        _WPREP = self._outer
        return False

    def begin_step(self, like):
        if not self.items:
            return
        for key, (_, w, _) in self.entries.items():
            if w.data_ptr() != key[1]:
                raise _lib.SprkError("WeightPrep: the storage of a %s weight was replaced after its transform was recorded; "
                                     "call clear() (or build a new stepper) after swapping parameters" % (tuple(w.shape),))
        if self._table is None or len(self._table) != len(self.items):
            self._table = (_lib.WprepItem * len(self.items))(*self.items)
        check(_lib.lib().sprk_prepare_weights(self._table, len(self.items), _stream(like)), "sprk_prepare_weights")
        self.launches += 1

    def lookup(self, key):
        """-> (workspace, dtype bits of a prepared call: SPRK_DT_WPREP + the kind the workspace holds) or (None, 0)"""
        e = self.entries.get(key)
        if e is None:
            return None, 0
        return e[0], _lib.DT_WPREP | ((e[2] & 7) << 12)

    def record(self, key, w, ws, item):
        self.entries[key] = (ws, w, int(item.kind))
        if item.kind:
            self.items.append(item)


_WPREP = None


def _prep_fwd(w, g, ep_key, ep):
    """-> (geometry to call with, workspace or None).  Inside a WeightPrep context: the kept workspace of this
    (weights, geometry, epilogue shape), with SPRK_DT_WPREP set once its transform is part of begin_step()."""
    if _WPREP is None:
        return g, None
    key = ("f", w.data_ptr(), tuple(geom_list(g)), ep_key)
    ws, bits = _WPREP.lookup(key)
    if ws is not None:
        gp = ConvGeom(*geom_list(g))
        gp.dtype = g.dtype | bits
        return gp, ws
    L = _lib.lib()
    ws = _ws(L.sprk_conv2d_fwd_ws_bytes(ctypes.byref(g)), w)
    item = _lib.WprepItem()
    check(L.sprk_conv2d_fwd_wprep(_p(w), ctypes.byref(g), ctypes.byref(ep), _p(ws), ws.numel(), ctypes.byref(item)),
          "sprk_conv2d_fwd_wprep")
    _WPREP.record(key, w, ws, item)
    return g, ws          # this first call still runs its own transform, into the kept workspace


def _prep_bwd(w, g):
    if _WPREP is None:
        return g, None
    key = ("b", w.data_ptr(), tuple(geom_list(g)))
    ws, bits = _WPREP.lookup(key)
    if ws is not None:
        gp = ConvGeom(*geom_list(g))
        gp.dtype = g.dtype | bits
        return gp, ws
    L = _lib.lib()
    ws = _ws(L.sprk_conv2d_bwd_data_ws_bytes(ctypes.byref(g)), w)
    item = _lib.WprepItem()
    check(L.sprk_conv2d_bwd_data_wprep(_p(w), ctypes.byref(g), _p(ws), ws.numel(), ctypes.byref(item)),
          "sprk_conv2d_bwd_data_wprep")
    _WPREP.record(key, w, ws, item)
    return g, ws


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
    _need_act(x, x2)
    _need_gpu(w, bias, scale, shift, res)
    ws = None
    if _WPREP is not None:
        ep = torch_ops._epilogue(bias, scale, shift, res, int(res_off), int(act), up_out)
        g, ws = _prep_fwd(w, g, (bool(up_out), res is not None), ep)
    return _S.conv2d_fwd(x, x2, w, bias, scale, shift, res, geom_list(g), int(res_off), int(act), 1 if up_out else 0, ws)


_F32 = torch.float32
_STORAGE = _lib.DT_X16 | _lib.DT_Y16
_CAP16 = {}


def storage16_caps(g, up_out=False):
    """Which calls of this layer have kernels for 16-bit ACTIVATION tensors (include/sprk.h, SPRK_DT_X16 / _Y16):
    bit 0 forward, bit 1 backward-data, bit 2 backward-weight.  0 for fp32 operands.  Cached per geometry."""
    if (g.dtype & 0xff) == 0:
        return 0
    fields = geom_list(g)
    key = tuple(fields[:15]) + (g.dtype & (0xff | _lib.DT_FORCE | _lib.DT_NAIVE), bool(up_out))
    caps = _CAP16.get(key)
    if caps is None:
        q = ConvGeom(*fields)
        q.dtype = g.dtype & (0xff | _lib.DT_FORCE | _lib.DT_NAIVE | _lib.DT_PIN)
        ep = ConvEpilogue(None, None, None, None, 0, 0, 0, ACT_NONE, 1 if up_out else 0)
        caps = int(_lib.lib().sprk_conv2d_storage16(ctypes.byref(q), ctypes.byref(ep)))
        _CAP16[key] = caps
    return caps


def _with_dtype(g, dtype):
    q = ConvGeom(*geom_list(g))
    q.dtype = dtype
    return q


class _Conv2dFn(torch.autograd.Function):
    """Storage types (round 4): with 16-bit operands (dtype bf16 / fp16) the activation tensors may themselves be 16-bit
    tensors.  A call whose kernel exists for them (storage16_caps) takes 16-bit inputs as they are and — store16 — writes
    a 16-bit output; any other call converts what it is handed to fp32 first, so every combination of layers works and
    the fast path is simply the one without conversions.  Gradients have the storage type of the tensor they belong to."""

    @staticmethod
    def forward(ctx, x, x2, w, bias, up1, stride, dil, pad, act, up_out, dtype, x_act=ACT_NONE, premasked=False,
                store16=False):
        ctx.x_act, ctx.premasked = x_act, premasked
        ctx.in_dtypes = (x.dtype, None if x2 is None else x2.dtype)
        x = x.contiguous()
        x2 = None if x2 is None else x2.contiguous()
        w = w.contiguous()
        g = make_geom(x, x2, w, up1, stride, dil, pad, dtype=dtype)
        c16 = dtype & 0xff
        caps = storage16_caps(g, up_out) if c16 else 0
        t16 = torch_ops.TORCH_OF[c16] if c16 else None
        in16 = x.dtype != _F32 or (x2 is not None and x2.dtype != _F32)
        if in16 and (caps & 1) and all(t is None or t.dtype in (_F32, t16) for t in (x, x2)):
            # the kernel reads 16-bit tensors: both sources in that type (a small fp32 source — the raw image — is cast)
            x = x if x.dtype == t16 else x.to(t16)
            x2 = x2 if x2 is None or x2.dtype == t16 else x2.to(t16)
        elif in16:
            x = x.float()
            x2 = None if x2 is None else x2.float()
            in16 = False
        g.dtype |= (_lib.DT_X16 if in16 else 0) | (_lib.DT_Y16 if (store16 and (caps & 1)) else 0)
        y = conv2d_forward(x, x2, w, g, bias=bias, act=act, up_out=up_out)
        ctx.geom = g
        ctx.caps = caps
        ctx.act = act
        ctx.up_out = up_out
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, x2, w, y if act != ACT_NONE else None, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, x2, w, y, bias = ctx.saved_tensors
        g = ctx.geom
        base = g.dtype & ~_STORAGE                    # operand type + FORCE / NAIVE bits, no storage bits
        c16 = base & 0xff
        t16 = torch_ops.TORCH_OF[c16] if c16 else None
        caps = ctx.caps
        _need_act(gy)
        # a gradient that is a channel slice of a concat layer's input gradient (dense planes, strided images) is read
        # in place by act_bwd; everything else wants it dense
        if not (ctx.act != ACT_NONE or ctx.up_out):
            gy = gy.contiguous()
        gx = gx2 = gw = gb = None
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        # gradient w.r.t. the pre-activation output (+ bias gradient)
        up2 = 1 if ctx.up_out else 0
        # premasked: the consumer of this layer's output (the next convolution's backward-data, or the pooling backward)
        # has already multiplied the gradient by act'(y): only the bias sum is left of the activation backward
        act_here = ACT_NONE if ctx.premasked else ctx.act
        if act_here != ACT_NONE or need_b or up2:
            want_gpre = act_here != ACT_NONE or bool(up2)
            defer_b = False
            if need_b:
                gb, defer_b = _grad_dest(bias)
            if not want_gpre:
                gy = gy.contiguous()
            out = _S.act_bwd(gy, y, act_here, [g.N, g.Cout, g.Hout, g.Wout], up2, want_gpre, gb, defer_b, torch_ops.code(gy))
            gpre = out if want_gpre else gy
        else:
            gpre = gy.contiguous()
        if ctx.needs_input_grad[2]:
            gw, defer_w = _grad_dest(w)
            xs, x2s, gs, gq = x, x2, gpre, _with_dtype(g, base)
            if any(t is not None and t.dtype != _F32 for t in (xs, x2s, gs)):
                if (caps & 4) and all(t is None or t.dtype in (_F32, t16) for t in (xs, x2s, gs)):
                    xs, gs = (t if t.dtype == t16 else t.to(t16) for t in (xs, gs))
                    x2s = x2s if x2s is None or x2s.dtype == t16 else x2s.to(t16)
                    gq.dtype = base | _lib.DT_X16
                else:       # no 16-bit-storage kernel for this layer's backward-weight: fp32 tensors
                    xs, gs = xs.float(), gs.float()
                    x2s = None if x2s is None else x2s.float()
            _S.conv2d_bwd_weight(xs, x2s, gs, geom_list(gq), gw, defer_w)
        need0 = ctx.needs_input_grad[0]
        need1 = x2 is not None and ctx.needs_input_grad[1]
        if need0 or need1:
            gd, wd = _with_dtype(g, base), w
            if g.C2 and not need1:
                # the skip source needs no gradient (the raw image x0 of decode_block_1): only the
                # first C1 input channels are back-propagated
                gd = ConvGeom(g.N, g.C1, 0, g.Hin, g.Win, g.up1, g.Cout, g.Hout, g.Wout, g.KH, g.KW, g.stride, g.dil,
                              g.pad_top, g.pad_left, base)
                wd = w[:, :g.C1].contiguous()
            gsrc = gpre
            if gd.stride > 1 and gd.C1 + gd.C2 >= 16 and not (gd.dtype & _lib.DT_NAIVE):
                # strided layers (the detector's down-sampling convolutions): the gradient w.r.t. the input is the
                # stride-1 backward-data of gy with stride - 1 zeros between its samples — the MFMA kernel on a
                # zero-stuffed copy (tiny tensors) instead of the direct kernel (which stays for the 7x7 stem: one
                # input channel would leave 15 of 16 MFMA columns empty)
                st = gd.stride
                H1, W1 = (gd.Hout - 1) * st + 1, (gd.Wout - 1) * st + 1
                gsrc = gpre.new_zeros((gd.N, gd.Cout, H1, W1))
                gsrc[:, :, ::st, ::st] = gpre
                gd = ConvGeom(gd.N, gd.C1, gd.C2, gd.Hin, gd.Win, gd.up1, gd.Cout, H1, W1, gd.KH, gd.KW, 1, gd.dil,
                              gd.pad_top, gd.pad_left, gd.dtype)
            # storage types of this call: the gradient it reads as it is (if a kernel exists for that), the gradient it
            # writes in the type of the tensor it belongs to
            want16 = t16 is not None and ctx.in_dtypes[0] == t16
            if (caps & 2) and gd.stride == 1 and gsrc.dtype in (_F32, t16):
                gd.dtype = base | (_lib.DT_X16 if gsrc.dtype == t16 else 0) | (_lib.DT_Y16 if want16 else 0)
            elif gsrc.dtype != _F32:
                gsrc = gsrc.float()
            # x_act: x is the output of an activated layer that this convolution alone consumes — its activation
            # backward is fused into this backward-data call (the saved input is the mask)
            masked = ctx.x_act != ACT_NONE and gd.C2 == 0 and not gd.up1
            ws_b = None
            if wd is w:      # (a sliced weight is a new tensor every step: nothing to keep)
                gd, ws_b = _prep_bwd(w, gd)
            mask = None
            if masked:
                mt = t16 if (gd.dtype & _lib.DT_Y16) else _F32
                mask = x if x.dtype == mt else x.to(mt)
            gin = _S.conv2d_bwd_data(gsrc, wd, geom_list(gd), mask, ctx.x_act if masked else ACT_NONE, ws_b)
            if ctx.x_act != ACT_NONE and not masked:
                raise _lib.SprkError("conv2d: x_act needs a single, full-resolution input source")
            if gd.C2 == 0 and not gd.up1:
                gx = gin
            elif not gd.up1:
                # concat without upsampling on load: the two halves are handed on as views of gin (their consumers —
                # act_bwd of the producing layers, autograd's accumulation — read strided images in place)
                gx, gx2 = gin[:, :gd.C1], gin[:, gd.C1:]
            else:
                gx, gx2 = _S.concat_up_bwd(gin.float() if gin.dtype != _F32 else gin, gd.C1, gd.C2, gd.up1, list(x.shape),
                                           list(x2.shape) if gd.C2 else [0])
                if not gd.C2:
                    gx2 = None
            # (a source that arrived in another storage type than the kernel's gets its gradient in ITS type)
            if gx is not None and gx.dtype != ctx.in_dtypes[0]:
                gx = gx.to(ctx.in_dtypes[0])
            if gx2 is not None and ctx.in_dtypes[1] is not None and gx2.dtype != ctx.in_dtypes[1]:
                gx2 = gx2.to(ctx.in_dtypes[1])
        return gx, gx2, gw, gb, None, None, None, None, None, None, None, None, None, None


def conv2d(x, w, bias=None, x2=None, up1=False, stride=1, dil=1, pad=(0, 0, 0, 0), act=ACT_NONE, up_out=False,
           dtype=0, x_act=ACT_NONE, premasked=False, store16=False):
    """y = act(conv(cat(up2(x) if up1 else x, x2), w) + bias); pad = (top, bottom, left, right).
    up_out: return nearest-x2-upsampled y (the upsampling is fused into the conv's stores).
    dtype: _lib.DT_F32 / DT_BF16 / DT_F16 — precision of the MFMA operands in forward, backward-data and
    backward-weight (a request: layers without a 16-bit kernel run in fp32; tensors are fp32 either way).
    Activation backward fused into the neighbours (conv -> conv and conv -> pool chains; a pair of promises the CALLER
    makes, networks.py): ``premasked`` — this layer's output is consumed by exactly one operator, which was told so
    (``x_act``) and returns the gradient already multiplied by act'(y); ``x_act`` — x is such an output."""
    _need_act(x, x2)
    _need_gpu(w, bias)
    if premasked and (up_out or act == ACT_NONE):
        raise ValueError("conv2d: premasked needs an activated, not upsampled output")
    return _Conv2dFn.apply(x, x2, w, bias, bool(up1), int(stride), int(dil), tuple(int(p) for p in pad), int(act),
                           bool(up_out), int(dtype), int(x_act), bool(premasked), bool(store16))


# ---- U-Net plumbing -----------------------------------------------------------------------------
class _ShiftMaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, shift, x_act):
        x = x.contiguous()
        _need_act(x)
        if x.shape[2] % 2 or x.shape[3] % 2:
            raise ValueError("shift_maxpool2: odd spatial size %dx%d" % (x.shape[2], x.shape[3]))
        ctx.shift, ctx.x_act = shift, x_act
        ctx.save_for_backward(x)
        return _S.shift_maxpool2_fwd(x, shift)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return _S.shift_maxpool2_bwd(gy.contiguous(), x, ctx.shift, ctx.x_act), None, None


def shift_maxpool2(x, shift=1, x_act=ACT_NONE):
    """x_act: x is the output of an activated convolution (called with premasked=True) that only this pool consumes:
    the gradient is returned multiplied by act'(x)."""
    return _ShiftMaxPoolFn.apply(x, int(shift), int(x_act))


class _Rot4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _need_gpu(x)
        if x.shape[2] != x.shape[3]:
            raise ValueError("rot4_stack needs square images, got %dx%d" % (x.shape[2], x.shape[3]))
        return _S.rot4_stack_fwd(x)

    @staticmethod
    def backward(ctx, gy):
        return _S.rot4_stack_bwd(gy.contiguous())


def rot4_stack(x):
    return _Rot4Fn.apply(x)


class _UnrotFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d):
        d = d.contiguous()
        _need_act(d)
        if d.shape[2] != d.shape[3] or d.shape[0] % 4:
            raise ValueError("unrot4_shift_concat: bad shape %s" % (tuple(d.shape),))
        return _S.unrot4_shift_concat_fwd(d)

    @staticmethod
    def backward(ctx, gf):
        return _S.unrot4_shift_concat_bwd(gf.contiguous())


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
        N = x.shape[0]
        if N % groups:
            raise ValueError("batch_norm_train: batch %d is not divisible into %d groups" % (N, groups))
        y, mean, invstd = _S.bn_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu, groups)
        ctx.relu, ctx.groups = relu, groups
        ctx.save_for_backward(x, y, gamma, mean, invstd, beta)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, mean, invstd, beta = ctx.saved_tensors
        gy = gy.contiguous()
        gg, gb = _grad_like(gamma), _grad_like(beta)     # straight into the parameters' gradient tensors
        gx = _S.bn_train_bwd(gy, x, y, gamma, mean, invstd, ctx.relu, ctx.groups, gg, gb)
        return gx, gg, gb, None, None, None, None, None, None


def batch_norm_train(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, relu=False, groups=1):
    return _BNTrainFn.apply(x, gamma, beta, running_mean, running_var, float(momentum), float(eps), bool(relu),
                            int(groups))


def batch_norm_eval(x, gamma, beta, running_mean, running_var, eps=1e-5, relu=False):
    """Inference-only (no autograd)."""
    x = x.contiguous()
    _need_gpu(x)
    return _S.bn_eval_fwd(x, gamma, beta, running_mean, running_var, float(eps), bool(relu))


# ---- per-pixel pipeline maths -------------------------------------------------------------------
class _ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out_stats, eps):
        out_stats = out_stats.contiguous()
        eps = eps.contiguous()
        _need_gpu(out_stats, eps)
        if out_stats.shape[1] != 2:
            raise ValueError("reparameterize expects out_stats with 2 channels")
        ctx.save_for_backward(out_stats, eps)
        return _S.reparam_fwd(out_stats, eps)

    @staticmethod
    def backward(ctx, gz):
        out_stats, eps = ctx.saved_tensors
        return _S.reparam_bwd(gz.contiguous(), out_stats, eps), None


def reparameterize(out_stats, eps):
    return _ReparamFn.apply(out_stats, eps)


class _SigmoidClampFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _need_gpu(x)
        ctx.save_for_backward(x)
        return _S.sigmoid_clamp_fwd(x)

    @staticmethod
    def backward(ctx, gp):
        (x,) = ctx.saved_tensors
        return _S.sigmoid_clamp_bwd(gp.contiguous(), x)


def sigmoid_clamp(x):
    return _SigmoidClampFn.apply(x)


class _PuLossFn(torch.autograd.Function):
    """PU detection loss (utils/losses.py:303-349); value and gradient come out of the same launch."""

    @staticmethod
    def forward(ctx, p, y, log_binom, slack):
        _need_gpu(p, y, log_binom)
        loss, gp = _S.pu_loss(p.contiguous().reshape(-1), y.contiguous().reshape(-1), log_binom.contiguous(), float(slack))
        ctx.save_for_backward(gp)
        ctx.p_shape = p.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        (gp,) = ctx.saved_tensors
        return (gp * gl).reshape(ctx.p_shape), None, None, None


def pu_loss(p, y, log_binom, slack=4.0):
    """p: scores in (0, 1); y: labels (>= 0 labelled, -1 unlabelled), fp32; log_binom: [B+1, B+1] table whose row N is
    binom.logpmf(0..N; N, tau)."""
    return _PuLossFn.apply(p, y, log_binom, slack)


# ---- fused pieces of the training step's tail (include/sprk.h, ABI 410) --------------------------------------------
class _CropAddFn(torch.autograd.Function):
    """out = y + x[:, :, off::stride, off::stride][:, :, :Ho, :Wo]   (y None: the crop alone)"""

    @staticmethod
    def forward(ctx, y, x, off, stride, out_h, out_w):
        x = x.contiguous()
        y = None if y is None else y.contiguous()
        _need_gpu(x, y)
        if x.dtype != _F32 or (y is not None and (y.dtype != _F32 or tuple(y.shape) != (x.shape[0], x.shape[1], out_h, out_w))):
            raise ValueError("crop_add: fp32 tensors, y of the cropped shape")
        ctx.geo = (off, stride, x.shape[2], x.shape[3], y is not None)
        return _S.crop_add_fwd(y, x, off, stride, out_h, out_w)

    @staticmethod
    def backward(ctx, g):
        off, stride, hx, wx, has_y = ctx.geo
        g = g.contiguous()
        gx = _S.crop_embed_bwd(g, off, stride, hx, wx) if ctx.needs_input_grad[1] else None
        return (g if has_y else None), gx, None, None, None, None


def crop_add(y, x, off, stride=1):
    """ResidA's residual (models/feature_extractor.py:403-411): y + x[:, :, off:-off, off:-off][:, :, ::stride, ::stride]
    in one launch, its gradient for x (zero-embedded) in one launch.  y None: the cropped copy alone."""
    hc, wc = x.shape[2] - 2 * off, x.shape[3] - 2 * off
    out_h, out_w = (hc + stride - 1) // stride, (wc + stride - 1) // stride
    return _CropAddFn.apply(y, x, int(off), int(stride), out_h, out_w)


class _NoiseStdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est):
        est = est.contiguous()
        _need_gpu(est)
        out, z = _S.noise_std_fwd(est)
        ctx.save_for_backward(z)
        ctx.shape = tuple(est.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        _, c, h, w = ctx.shape
        return _S.noise_std_bwd(g.reshape(-1).contiguous(), z, c, h, w)


NOISE_STD_MAX_PIXELS = 1 << 16


def noise_std_from_map(est):
    """softplus(mean(est, (1,2,3)) - 4) + 1e-3 per image -> [B,1,1,1] (denoiser_v2.py:392-402), one launch each way.
    One workgroup per image: for patches; callers keep torch's reduction for whole micrographs (NOISE_STD_MAX_PIXELS)."""
    return _NoiseStdFn.apply(est)


class _JointLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss_out, pred, p, pf, axis, alpha, wc):
        pshape = tuple(pred.shape)
        loss_out, pred = loss_out.contiguous(), pred.reshape(1).contiguous()
        p, pf = p.contiguous(), pf.contiguous()
        _need_gpu(loss_out, pred, p, pf)
        if p.dim() != 4 or p.shape[1] != 1 or p.shape != pf.shape or loss_out.numel() != p.shape[0]:
            raise ValueError("joint_loss: scores [B,1,H,W] twice and one denoising loss per image")
        final, consis = _S.joint_loss_fwd(loss_out, pred, p, pf, axis, float(alpha), float(wc))
        ctx.save_for_backward(p, pf)
        ctx.cfg = (axis, float(alpha), float(wc), tuple(loss_out.shape), pshape)
        ctx.mark_non_differentiable(consis)
        return final.reshape(loss_out.shape), consis.reshape(())

    @staticmethod
    def backward(ctx, g, _gc):
        p, pf = ctx.saved_tensors
        axis, alpha, wc, lshape, pshape = ctx.cfg
        gl, gpred, gp, gpf = _S.joint_loss_bwd(g.reshape(-1).contiguous(), p, pf, axis, alpha, wc)
        return gl.reshape(lshape), gpred.reshape(pshape), gp, gpf, None, None, None


def joint_loss(loss_out, pred_loss, p, pf_unflipped, axis, alpha, w_consis=0.1):
    """(final [B,1], consis []) of the joint training step (denoiser_v2.py:516-519):
    consis = mse(p, flip(pf, axis)), final = alpha * loss_out + (1 - alpha) * pred_loss + w_consis * consis.
    axis: -1 / 3 (W) or -2 / 2 (H).  consis is returned for the logs only (its gradient flows through final)."""
    ax = {-1: 0, 3: 0, -2: 1, 2: 1}[int(axis)]
    return _JointLossFn.apply(loss_out, pred_loss, p, pf_unflipped, ax, alpha, w_consis)


NOISE_GAUSSIAN, NOISE_POISSON = 0, 1      # SPRK_NOISE_* (include/sprk.h)


class _SsdnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_stats, noise_std, style):
        x = x.contiguous()
        out_stats = out_stats.contiguous()
        ns = noise_std.reshape(-1).contiguous()
        _need_gpu(x, out_stats, ns)
        loss, pme, mstd, nsmap = _S.ssdn_fwd(x, out_stats, ns, style)
        ctx.save_for_backward(x, out_stats, ns)
        ctx.ns_shape = noise_std.shape
        ctx.style = style
        ctx.mark_non_differentiable(pme, mstd, nsmap)
        return loss, pme, mstd, nsmap

    @staticmethod
    def backward(ctx, gloss, _gpme, _gmstd, _gmap):
        x, out_stats, ns = ctx.saved_tensors
        go, gns = _S.ssdn_bwd(gloss.reshape(-1).contiguous(), x, out_stats, ns, ctx.style)
        return None, go, gns.reshape(ctx.ns_shape), None


def ssdn_nll_pme(x, out_stats, noise_std, style=NOISE_GAUSSIAN):
    """(loss [B,1] = per-image mean NLL, pme [B,1,H,W], model_std [1,B,H,W], noise_std_map).
    noise_std: the remapped estimate [B,1,1,1]; gaussian: it is the noise standard deviation and noise_std_map is
    empty; poisson (denoiser_v2.py:412-424): std = sqrt(max(mu, 1e-3) * estimate) per pixel -> noise_std_map [B,H,W]."""
    return _SsdnFn.apply(x, out_stats, noise_std, int(style))


def head1x1_eligible(f, c1, c2, c3):
    """The fused inference head (sprk_head1x1_fwd) covers 384 -> 384 -> 96 -> {1, 2} and 96 -> 96 -> 96 -> {1, 2} with
    LeakyReLU(0.1), fp32 operands, planes whose pixel count is a multiple of 128."""
    K0, N1 = c1.weight.shape[1], c1.weight.shape[0]
    return (not torch.is_grad_enabled() and f.is_cuda and f.dtype == torch.float32 and (K0, N1) in ((384, 384), (96, 96))
            and tuple(c2.weight.shape[:2]) == (96, N1) and c3.weight.shape[1] == 96 and c3.weight.shape[0] in (1, 2)
            and c1.kernel_size == (1, 1) and c1.act == ACT_LEAKY and c2.act == ACT_LEAKY and c3.act == ACT_NONE
            and all(c.bias is not None and (c.mfma_dtype_nograd & 0xff) == 0 for c in (c1, c2, c3))
            and (f.shape[2] * f.shape[3]) % 128 == 0 and f.shape[2] * f.shape[3] < (1 << 25) and f.shape[1] == K0)


def head1x1(f, c1, c2, c3):
    """out = c3(c2(c1(f))) for three 1x1 convolution modules, one launch (inference only, no autograd)."""
    f = f.contiguous()
    _need_gpu(f)
    return _S.head1x1_fwd(f, c1.weight.contiguous(), c1.bias, c2.weight.contiguous(), c2.bias, c3.weight.contiguous(), c3.bias)


def head1x1_unrot_eligible(d, c1, c2, c3):
    """Blind-spot head straight from the rotated stack d [4B,96,P,P] (no [B,384,P,P] tensor): sprk_head1x1_unrot_fwd."""
    if d.dim() != 4 or d.shape[0] % 4 or d.shape[1] != 96 or d.shape[2] != d.shape[3] or d.shape[2] % 16:
        return False
    K0, N1 = c1.weight.shape[1], c1.weight.shape[0]
    return (not torch.is_grad_enabled() and d.is_cuda and d.dtype == torch.float32 and (K0, N1) == (384, 384)
            and tuple(c2.weight.shape[:2]) == (96, 384) and c3.weight.shape[1] == 96 and c3.weight.shape[0] in (1, 2)
            and c1.kernel_size == (1, 1) and c1.act == ACT_LEAKY and c2.act == ACT_LEAKY and c3.act == ACT_NONE
            and all(c.bias is not None and (c.mfma_dtype_nograd & 0xff) == 0 for c in (c1, c2, c3))
            and d.shape[2] * d.shape[3] < (1 << 25))


def head1x1_unrot(d, c1, c2, c3):
    d = d.contiguous()
    _need_gpu(d)
    return _S.head1x1_unrot_fwd(d, c1.weight.contiguous(), c1.bias, c2.weight.contiguous(), c2.bias, c3.weight.contiguous(),
                                c3.bias)
