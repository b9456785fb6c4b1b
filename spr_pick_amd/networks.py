"""Host-side mirror of the reference's network classes, running on libsprk.so.

Same class names, constructor arguments, attribute / parameter names (hence the same
``state_dict`` keys, so reference checkpoints load unchanged) and forward semantics as

  spr_pick/models/joint_network_v2.py   DualNetwork :11-286, JointNetwork :437-532,
                                        Detector :543-561, ShiftConv2d :565-584
  spr_pick/models/joint_network_v2_shallow.py  DualNetworkShallow :11-307
  spr_pick/models/feature_extractor.py  ResNet :12-61, ResNet8 :102-144, BasicConv2d :279-324,
                                        ResidA :326-416
  spr_pick/models/classifier.py         LinearClassifier :7-38
  spr_pick/models/utility.py            Shift2d :46-72

but the computation is re-organised for the GPU: convolution + bias + activation is one
kernel (the ``LeakyReLU`` slots of the reference's ``nn.Sequential`` blocks are kept as inert
markers so indices match), pad/crop of ShiftConv2d is index arithmetic inside the conv,
upsample + concat is fused into the following conv's loads, ``fill()``/``unfill()`` flip a flag
instead of rewriting module attributes, and eval-mode BatchNorm is folded into the conv epilogue.
"""
import math
import os

import torch
import torch.nn as nn

from . import ops
from ._lib import DT_PIN
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU


class FusedAct(nn.Module):
    """Marker occupying the activation slot of a reference ``nn.Sequential``: the activation
    itself is applied inside the preceding convolution kernel."""

    def __init__(self, kind="LeakyReLU(0.1)"):
        super().__init__()
        self.kind = kind

    def extra_repr(self):
        return self.kind + ", fused into the preceding conv"

    def forward(self, x):
        return x


class Conv2d(nn.Module):
    """nn.Conv2d counterpart (square kernel, symmetric zero padding) with a fused activation."""

    shifted = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True,
                 act=ACT_NONE):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.act = act
        self.mfma_dtype = 0      # _lib.DT_*: operand precision of this layer's MFMA kernels (set_conv_dtype) ...
        self.mfma_dtype_nograd = 0   # ... while autograd records (training) / under no_grad (inference)
        self.store16 = False         # 16-bit modes: write the output as a bf16 / fp16 tensor where a kernel exists (ops.conv2d)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):  # torch.nn.Conv2d defaults
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def _pad(self):
        p = self.padding
        return (p, p, p, p)

    def forward(self, x, skip=None, up=False, up_out=False, x_act=ACT_NONE, premasked=False):
        """skip: second input source (channel concat); up: x is half resolution, upsampled on load;
        up_out: write the output nearest-upsampled x2 (the nn.Upsample that follows in the reference);
        x_act / premasked: activation backward fused into the neighbouring operators (ops.conv2d)."""
        # inference (no autograd): the kernel choice is pinned to the layer's structure (_lib.DT_PIN), so that a window
        # of a micrograph gets the very arithmetic the whole micrograph gets (Denoiser._tiled_networks)
        dtype = self.mfma_dtype if torch.is_grad_enabled() else (self.mfma_dtype_nograd | DT_PIN)
        return ops.conv2d(x, self.weight, self.bias, x2=skip, up1=up, stride=self.stride, dil=self.dilation,
                          pad=self._pad(), act=self.act, up_out=up_out, dtype=dtype, x_act=x_act, premasked=premasked,
                          store16=self.store16 and (dtype & 0xff) != 0)

    def extra_repr(self):
        return "%d, %d, kernel_size=%s, stride=%d, padding=%d, dilation=%d, act=%d%s" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.dilation, self.act,
            ", shifted" if self.shifted else "")


class ShiftConv2d(Conv2d):
    """Convolution whose receptive field lies at/above the output row (Laine et al.): the
    reference pads k = kh//2 rows on top, convolves with ``padding``, crops the last k rows;
    here that is simply top padding 2k, bottom padding 0."""

    shifted = True

    def _pad(self):
        k = self.kernel_size[0] // 2
        p = self.padding
        return (p + k, p - k, p, p)


class Shift2d(nn.Module):
    """Shift2d((1,0)): kept for structural parity; the shift is fused into the pooling /
    un-rotation kernels."""

    def __init__(self, shift):
        super().__init__()
        self.shift = shift

    def forward(self, x):
        raise RuntimeError("Shift2d is fused into its consumer kernel and is never called on its own")


class ShiftMaxPool(nn.Module):
    def __init__(self, shift):
        super().__init__()
        self.shift = shift

    def forward(self, x, x_act=ACT_NONE):
        return ops.shift_maxpool2(x, self.shift, x_act)


FUSED_UNROT = True      # debug: False = materialise the un-rotated [B,384,P,P] tensor in front of the fused head
FUSED_HEAD = True       # debug: False = the three 1x1 convolutions of a U-Net's head as separate launches in inference too
FUSE_ACT_BWD = True     # debug: False = every convolution runs its own activation-backward pass


def _kaiming_leaky(mod):
    for m in mod.modules():
        if isinstance(m, Conv2d):
            nn.init.kaiming_normal_(m.weight.data, a=0.1)
            if m.bias is not None:
                m.bias.data.zero_()


class _UNetBase(nn.Module):
    """Shared construction of the two U-Nets (48-channel encoder, 96-channel decoder)."""

    def _conv(self, cin, cout, k, act=ACT_LEAKY):
        cls = ShiftConv2d if self._blindspot else Conv2d
        return cls(cin, cout, k, stride=1, padding=k // 2, act=act)

    def _pool(self):
        if self._blindspot:
            return nn.Sequential(Shift2d((1, 0)), ShiftMaxPool(1))
        return ShiftMaxPool(0)

    def _enc1(self, cin):
        return nn.Sequential(self._conv(cin, 48, 3), FusedAct(), self._conv(48, 48, 3), FusedAct(), self._pool())

    def _enc(self, pool=True):
        mods = [self._conv(48, 48, 3), FusedAct()]
        if pool:
            mods.append(self._pool())
        return nn.Sequential(*mods)

    def _dec(self, cin, up=True):
        mods = [self._conv(cin, 96, 3), FusedAct(), self._conv(96, 96, 3), FusedAct()]
        if up:
            mods.append(FusedAct("Upsample(x2, nearest)"))
        return nn.Sequential(*mods)

    @staticmethod
    def _run_pool(block, t, x_act=ACT_NONE):
        p = block[-1]
        return (p[1] if isinstance(p, nn.Sequential) else p)(t, x_act)

    # Activation backward fused into the neighbours (ops.conv2d: premasked / x_act).  Every conv -> conv and conv -> pool
    # link of the U-Nets qualifies (the intermediate tensor has exactly one consumer); the convolution side is taken
    # where the consumer's backward-data kernel applies the mask in its epilogue.
    @staticmethod
    def _fuse_chain(first, second):
        """second(first(x)): may the LeakyReLU backward of `first` run inside `second`'s backward-data kernel?"""
        return (torch.is_grad_enabled() and FUSE_ACT_BWD and first.act != ACT_NONE and second.kernel_size[0] == 3
                and second.mfma_dtype == 0)

    def _chain(self, first, second, x, skip=None, up_out=False):
        """second(first(x, skip)).  (Callers in the decoders write the two calls out, rebinding their variable, so that
        the block's input is released before the second convolution allocates its output: three 25.8 GB tensors
        instead of four at 4096^2.)"""
        fuse = self._fuse_chain(first, second)
        t = first(x, skip=skip, premasked=fuse)
        return second(t, up_out=up_out, x_act=first.act if fuse else ACT_NONE)

    def _conv_pool(self, conv, block, x):
        """pool(conv(x)): the LeakyReLU backward of `conv` runs inside the pooling backward."""
        fuse = torch.is_grad_enabled() and FUSE_ACT_BWD and conv.act != ACT_NONE
        return self._run_pool(block, conv(x, premasked=fuse), conv.act if fuse else ACT_NONE)

    def _head(self, t):
        """output_block (two 1x1 convolutions) + output_conv: in inference one fused launch (ops.head1x1)."""
        c1, c2, c3 = self.output_block[0], self.output_block[2], self.output_conv
        if FUSED_HEAD and ops.head1x1_eligible(t, c1, c2, c3):
            return ops.head1x1(t, c1, c2, c3)
        return c3(c2(c1(t)))

    @property
    def blindspot(self):
        return self._blindspot

    def init_weights(self):
        with torch.no_grad():
            _kaiming_leaky(self)
            if self._zero_output_weights:
                self.output_conv.weight.zero_()
            else:
                nn.init.kaiming_normal_(self.output_conv.weight.data, nonlinearity="linear")


def set_conv_dtype(module, dtype):
    """Operand precision ("f32" | "bf16" | "f16" | "mixed16") of the convolutions of every U-Net under ``module`` — the
    layers that carry 99.6 % of the FLOPs.  The detector (BatchNorm statistics over 64-patch
    batches, 0.4 % of the FLOPs) and each U-Net's final 1x1 output convolution (it produces mu and the variance factor
    the likelihood is evaluated on) always run in fp32.  Returns the number of layers switched.

    "mixed16" is the 16-bit mode of BASELINE configs[4]: bf16 operands while autograd records (forward, backward-data,
    backward-weight of a training step), fp16 operands under no_grad (inference).  Same MFMA rate; fp16 has 8x the
    resolution, which the picks want (agreement with fp32 picks 0.99), but its range loses the U-Nets'
    gradients: a run trained with fp16 operands end to end leaves the sigma-net at its floor and reaches AP 0.65 where
    fp32 and bf16 reach 0.93 (profiles/r04_full_pipeline.json)."""
    from ._lib import DT_BF16, DT_F16, DTYPES
    storage = os.environ.get("SPRK_STORE16", "1") != "0"
    if isinstance(dtype, str) and dtype.endswith("/operands"):       # "bf16/operands": fp32 tensors, 16-bit operands only
        dtype, storage = dtype[:-len("/operands")], False
    if dtype == "mixed16":
        code, code_ng = DT_BF16, DT_F16
    else:
        code = code_ng = DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
    n = 0
    for net in module.modules():
        if isinstance(net, _UNetBase):
            for name, m in net.named_modules():
                if isinstance(m, Conv2d) and not name.startswith("output_conv"):
                    m.mfma_dtype, m.mfma_dtype_nograd = code, code_ng
                    # 16-bit ACTIVATION tensors between the U-Net's layers (round 4): every layer stores its output in the
                    # operand type, except the one in front of the fp32 output convolution (output_block.2)
                    m.store16 = storage and (code & 0xff) != 0 and name != "output_block.2"
                    n += 1
    return n


class DualNetwork(_UNetBase):
    """Blind-spot (or plain) 5-level U-Net of the joint model."""

    def __init__(self, in_channels=1, out_channels=1, blindspot=False, detect=False, detect_out_channels=1,
                 zero_output_weights=False):
        super().__init__()
        self._blindspot = blindspot
        self._zero_output_weights = zero_output_weights
        self.detect = detect
        self.encode_block_1 = self._enc1(in_channels)
        self.encode_block_2 = self._enc()
        self.encode_block_3 = self._enc()
        self.encode_block_4 = self._enc()
        self.encode_block_5 = self._enc()
        self.encode_block_6 = self._enc(pool=False)
        self.decode_block_6 = nn.Sequential(FusedAct("Upsample(x2, nearest)"))
        self.decode_block_5 = self._dec(96)
        self.decode_block_4 = self._dec(144)
        self.decode_block_3 = self._dec(144)
        self.decode_block_2 = self._dec(144)
        self.decode_block_1 = self._dec(96 + in_channels, up=False)
        if blindspot:
            self.shift = Shift2d((1, 0))
            nin = 384
        else:
            nin = 96
        self.output_block = nn.Sequential(self._conv(nin, nin, 1), FusedAct(), self._conv(nin, 96, 1), FusedAct())
        if detect:
            self.output_conv_f = self._conv(96, 1, 1, act=ACT_NONE)
        self.output_conv = self._conv(96, out_channels, 1, act=ACT_NONE)
        self.init_weights()

    def forward(self, x):
        if self._blindspot:
            if x.shape[-1] != x.shape[-2]:
                raise ValueError("blind-spot network needs square inputs (4-rotation stack), got %s" % (tuple(x.shape),))
            x = ops.rot4_stack(x)
        e1, e2, e3, e4, e5, e6 = (self.encode_block_1, self.encode_block_2, self.encode_block_3, self.encode_block_4,
                                  self.encode_block_5, self.encode_block_6)
        fuse = torch.is_grad_enabled() and FUSE_ACT_BWD
        t = e1[0](x, premasked=fuse and e1[2].mfma_dtype == 0)
        # encode_block_1 is conv -> conv -> pool
        t = e1[2](t, x_act=ACT_LEAKY if fuse and e1[2].mfma_dtype == 0 else ACT_NONE, premasked=fuse)
        pool1 = self._run_pool(e1, t, ACT_LEAKY if fuse else ACT_NONE)
        pool2 = self._conv_pool(e2[0], e2, pool1)
        pool3 = self._conv_pool(e3[0], e3, pool2)
        pool4 = self._conv_pool(e4[0], e4, pool3)
        pool5 = self._conv_pool(e5[0], e5, pool4)
        # every nn.Upsample of the reference is fused into the stores of the conv that feeds it
        t = e6[0](pool5, up_out=True)
        for blk, skip, up_out in ((self.decode_block_5, pool4, True), (self.decode_block_4, pool3, True),
                                  (self.decode_block_3, pool2, True), (self.decode_block_2, pool1, True),
                                  (self.decode_block_1, x, False)):
            fuse_c = self._fuse_chain(blk[0], blk[2])
            t = blk[0](t, skip=skip, premasked=fuse_c)
            t = blk[2](t, up_out=up_out, x_act=blk[0].act if fuse_c else ACT_NONE)
        if self._blindspot:
            c1, c2, c3 = self.output_block[0], self.output_block[2], self.output_conv
            if FUSED_HEAD and FUSED_UNROT and ops.head1x1_unrot_eligible(t, c1, c2, c3):
                # inference: Shift2d + un-rotation + concat are the input gather of the fused head
                out = ops.head1x1_unrot(t, c1, c2, c3)
                return (out, None) if self.detect else out
            t = ops.unrot4_shift_concat(t)
        out = self._head(t)
        if self._blindspot and self.detect:
            return out, None
        return out

    @staticmethod
    def input_wh_mul():
        return 2 ** 5


class DualNetworkShallow(_UNetBase):
    """3-level U-Net used as the per-image noise-sigma estimator (blindspot=False in the joint
    pipeline, denoiser_v2.py:129-137).  ``decode_block_3``, ``detect_block`` and
    ``output_conv_f`` are constructed, as in the reference, but unused by this configuration."""

    def __init__(self, in_channels=1, out_channels=1, blindspot=False, detect=False, detect_out_channels=1,
                 zero_output_weights=False):
        super().__init__()
        self._blindspot = blindspot
        self._zero_output_weights = zero_output_weights
        self.detect = detect
        self.encode_block_1 = self._enc1(in_channels)
        self.encode_block_2 = self._enc()
        self.encode_block_3 = self._enc()
        self.encode_block_6 = self._enc(pool=False)
        self.decode_block_6 = nn.Sequential(FusedAct("Upsample(x2, nearest)"))
        self.decode_block_5 = self._dec(96)
        self.decode_block_3 = self._dec(144)
        self.decode_block_2 = self._dec(144)
        self.decode_block_1 = self._dec(96 + in_channels, up=False)
        if blindspot:
            self.shift = Shift2d((1, 0))
            nin = 384
        else:
            nin = 96
        self.output_block = nn.Sequential(self._conv(nin, nin, 1), FusedAct(), self._conv(nin, 96, 1), FusedAct())
        self.detect_block = nn.Sequential(self._conv(nin, nin, 1), FusedAct(), self._conv(nin, 96, 1), FusedAct())
        self.output_conv = self._conv(96, out_channels, 1, act=ACT_NONE)
        self.output_conv_f = Conv2d(96, 1, 1)
        self.init_weights()

    def forward(self, x):
        if self._blindspot:
            raise NotImplementedError("DualNetworkShallow(blindspot=True) is not on the joint pipeline's path "
                                      "(reference: denoiser_v2.py:129-137 always passes blindspot=False)")
        e1, e2, e3, e6 = self.encode_block_1, self.encode_block_2, self.encode_block_3, self.encode_block_6
        fuse = torch.is_grad_enabled() and FUSE_ACT_BWD
        t = e1[0](x, premasked=fuse and e1[2].mfma_dtype == 0)
        t = e1[2](t, x_act=ACT_LEAKY if fuse and e1[2].mfma_dtype == 0 else ACT_NONE, premasked=fuse)
        pool1 = self._run_pool(e1, t, ACT_LEAKY if fuse else ACT_NONE)
        pool2 = self._conv_pool(e2[0], e2, pool1)
        pool3 = self._conv_pool(e3[0], e3, pool2)
        t = e6[0](pool3, up_out=True)
        for blk, skip, up_out in ((self.decode_block_5, pool2, True), (self.decode_block_2, pool1, True),
                                  (self.decode_block_1, x, False)):
            fuse_c = self._fuse_chain(blk[0], blk[2])
            t = blk[0](t, skip=skip, premasked=fuse_c)
            t = blk[2](t, up_out=up_out, x_act=blk[0].act if fuse_c else ACT_NONE)
        return self._head(t)

    @staticmethod
    def input_wh_mul():
        return 2 ** 3


# ---- detector: BatchNorm2d(1) + LinearClassifier(ResNet8(bn=True)) -----------------------------------
class BatchNorm2d(nn.Module):
    """Parameter/buffer holder with nn.BatchNorm2d's names; the maths runs in sprk kernels."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.groups = 1   # > 1: the batch holds that many stacked passes (JointNetwork.forward_pair)
        self.count_outside = False   # the caller advances num_batches_tracked of all its layers in one launch

    def forward(self, x, relu=False):
        if self.training:
            if not self.count_outside:
                self.num_batches_tracked += self.groups
            return ops.batch_norm_train(x, self.weight, self.bias, self.running_mean, self.running_var,
                                        self.momentum, self.eps, relu, groups=self.groups)
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            scale, shift = self.folded()
            y = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            return torch.relu(y) if relu else y
        return ops.batch_norm_eval(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, relu)

    def folded(self):
        """Eval-mode affine (scale, shift) for fusing into a conv epilogue."""
        scale = self.weight * torch.rsqrt(self.running_var + self.eps)
        return scale, self.bias - self.running_mean * scale


def _infer_mode():
    return not torch.is_grad_enabled()


class BasicConv2d(nn.Module):
    """conv (valid) -> BN -> ReLU.  ``fill(stride)`` turns the stride into dilation."""

    def __init__(self, nin, nout, kernel_size, dilation=1, stride=1, bn=False, activation=None):
        super().__init__()
        self.conv = Conv2d(nin, nout, kernel_size, dilation=dilation, stride=stride, bias=not bn)
        if bn:
            self.bn = BatchNorm2d(nout)
        self.act = FusedAct("ReLU")
        self.kernel_size, self.stride, self.dilation, self.og_dilation, self.padding = kernel_size, stride, dilation, dilation, 0
        self._is_filled = False

    def fill(self, stride):
        self._is_filled = True
        self.dilation = self.og_dilation * stride
        return self.stride

    def unfill(self):
        self._is_filled = False
        self.dilation = self.og_dilation

    def forward(self, x, pad=0):
        dil = self.dilation
        stride = 1 if self._is_filled else self.stride
        has_bn = hasattr(self, "bn")
        if has_bn and not self.training and _infer_mode():
            g = ops.make_geom(x, None, self.conv.weight, False, stride, dil, (pad, pad, pad, pad))
            scale, shift = self.bn.folded()
            return ops.conv2d_forward(x.contiguous(), None, self.conv.weight, g, act=ACT_RELU, scale=scale, shift=shift)
        y = ops.conv2d(x, self.conv.weight, self.conv.bias, stride=stride, dil=dil, pad=(pad, pad, pad, pad),
                       act=ACT_NONE if has_bn else ACT_RELU)
        return self.bn(y, relu=True) if has_bn else y


class ResidA(nn.Module):
    """t = ReLU(BN0(conv0(x))); y = conv1(t) [dilated, maybe strided]; skip = centre-crop(x)
    [1x1 proj if nin != nout]; out = ReLU(BN1(y + skip)) — BN after the residual add."""

    def __init__(self, nin, nhidden, nout, dilation=1, stride=1, activation=None, bn=False):
        super().__init__()
        self.bn = bn
        if nin != nout:
            self.proj = Conv2d(nin, nout, 1, stride=stride, bias=False)
        self.conv0 = Conv2d(nin, nhidden, 3, bias=not bn)
        if bn:
            self.bn0 = BatchNorm2d(nhidden)
        self.act0 = FusedAct("ReLU")
        self.conv1 = Conv2d(nhidden, nout, 3, dilation=dilation, stride=stride, bias=not bn)
        if bn:
            self.bn1 = BatchNorm2d(nout)
        self.act1 = FusedAct("ReLU")
        self.kernel_size = 2 * dilation + 3
        self.stride, self.dilation, self.padding = stride, 1, 0
        self.og_dilation1 = dilation
        self._is_filled = False

    def fill(self, stride):
        self._is_filled = True
        self.dilation = self.dilation * stride
        return self.stride

    def unfill(self):
        self._is_filled = False
        self.dilation = 1

    def forward(self, x):
        d0 = self.dilation
        d1 = self.og_dilation1 * self.dilation
        s = 1 if self._is_filled else self.stride
        e = d0 + d1
        infer = self.bn and not self.training and _infer_mode()
        has_proj = hasattr(self, "proj")
        if infer:
            x = x.contiguous()
            sc0, sh0 = self.bn0.folded()
            g0 = ops.make_geom(x, None, self.conv0.weight, False, 1, d0, (0, 0, 0, 0))
            t = ops.conv2d_forward(x, None, self.conv0.weight, g0, act=ACT_RELU, scale=sc0, shift=sh0)
            sc1, sh1 = self.bn1.folded()
            g1 = ops.make_geom(t, None, self.conv1.weight, False, s, d1, (0, 0, 0, 0))
            if has_proj or s > 1:
                skip = x[:, :, e:-e, e:-e]
                if has_proj:
                    skip = skip.contiguous()
                    gp = ops.make_geom(skip, None, self.proj.weight, False, s, 1, (0, 0, 0, 0))
                    skip = ops.conv2d_forward(skip, None, self.proj.weight, gp)
                else:
                    skip = skip[:, :, ::s, ::s].contiguous()
                return ops.conv2d_forward(t, None, self.conv1.weight, g1, act=ACT_RELU, scale=sc1, shift=sh1,
                                          res=skip, res_off=0)
            return ops.conv2d_forward(t, None, self.conv1.weight, g1, act=ACT_RELU, scale=sc1, shift=sh1,
                                      res=x, res_off=e)
        t = ops.conv2d(x, self.conv0.weight, self.conv0.bias, dil=d0, act=ACT_NONE if self.bn else ACT_RELU)
        if self.bn:
            t = self.bn0(t, relu=True)
        y = ops.conv2d(t, self.conv1.weight, self.conv1.bias, stride=s, dil=d1)
        if x.dtype == torch.float32 and y.dtype == torch.float32:
            # crop (and stride) + add in one launch, the crop's gradient in one (slicing costs autograd two fills and
            # two copies per crop on the way back)
            if has_proj:
                y = y + ops.conv2d(ops.crop_add(None, x, e), self.proj.weight, None, stride=s)
            else:
                y = ops.crop_add(y, x, e, s)
        else:
            skip = x[:, :, e:-e, e:-e]
            if has_proj:
                skip = ops.conv2d(skip.contiguous(), self.proj.weight, None, stride=s)
            elif s > 1:
                skip = skip[:, :, ::s, ::s]
            y = y + skip
        if self.bn:
            return self.bn1(y, relu=True)
        return torch.relu(y)


class ResNet8(nn.Module):
    """Topaz-style 8-layer residual feature extractor; receptive field 63 px."""

    def __init__(self, units=(32, 64, 128), bn=True, **kwargs):
        super().__init__()
        units = list(units)
        self.num_features = self.latent_dim = units[-1]
        self.stride = 2
        self.features = nn.Sequential(
            BasicConv2d(1, units[0], 7, stride=2, bn=bn),
            ResidA(units[0], units[0], units[0], dilation=2, bn=bn),
            ResidA(units[0], units[0], units[1], dilation=2, stride=2, bn=bn),
            ResidA(units[1], units[1], units[1], dilation=2, bn=bn),
            BasicConv2d(units[1], units[2], 3, bn=bn),
        )
        self.width = self._receptive_field()
        self.pad = False

    def _receptive_field(self):
        # walk the stack backwards from one output pixel (utils/utils.py:18-47 applied to this list)
        size = 1
        for m in reversed(list(self.features)):
            size = (size - 1) * m.stride + 1 + (m.kernel_size - 1) * m.dilation - 2 * m.padding
        return size

    def fill(self, stride=1):
        for mod in self.features.children():
            stride *= mod.fill(stride)
        self.pad = True
        return stride

    def unfill(self):
        for mod in self.features.children():
            mod.unfill()
        self.pad = False

    def forward(self, x):
        if x.dim() < 4:
            x = x.unsqueeze(1)
        f = self.features
        # filled: the reference zero-pads width//2 on every side before the first conv; here the
        # first conv reads those zeros through its own out-of-bounds handling
        h = f[0](x, pad=self.width // 2 if self.pad else 0)
        for m in (f[1], f[2], f[3], f[4]):
            h = m(h)
        return h


class LinearClassifier(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.features = features
        self.classifier = Conv2d(features.latent_dim, 1, 1)

    @property
    def width(self):
        return self.features.width

    @property
    def latent_dim(self):
        return self.features.latent_dim

    def fill(self, stride=1):
        return self.features.fill(stride=stride)

    def unfill(self):
        self.features.unfill()

    def forward(self, x):
        return self.classifier(self.features(x))


class Detector(nn.Module):
    def __init__(self):
        super().__init__()
        self.detector = LinearClassifier(ResNet8(bn=True))
        self.m = BatchNorm2d(1)

    def fill(self, stride=1):
        return self.detector.fill(stride=stride)

    def unfill(self):
        return self.detector.unfill()

    def forward(self, x):
        return self.detector(self.m(x))


class JointNetwork(nn.Module):
    """Blind-spot denoiser + reparameterised sample + per-pixel detector."""

    def __init__(self, in_channels=1, out_channels=1, blindspot=False, detect=False, detect_out_channels=1,
                 zero_output_weights=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.denoise_branch = DualNetwork(in_channels=in_channels, out_channels=out_channels, blindspot=blindspot,
                                          detect=detect, detect_out_channels=detect_out_channels,
                                          zero_output_weights=zero_output_weights)
        self.detector = Detector()

    def fill(self, stride=1):
        return self.detector.fill(stride=stride)

    def unfill(self):
        return self.detector.unfill()

    def reparameterize(self, x, eps=None):
        """z = mu + eps * A^2; eps ~ N(0,1) is drawn on the device unless given (the reference
        draws it with torch.randn_like in train AND eval, joint_network_v2.py:473)."""
        if self.in_channels != 1 or self.out_channels != 2:
            raise NotImplementedError("reparameterize: only the 1-channel (mu, A) layout is on the hot path")
        if eps is None:
            eps = torch.randn((x.shape[0], 1, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device)
        return ops.reparameterize(x, eps)

    @staticmethod
    def input_wh_mul():
        return 2 ** 5

    def forward(self, x, eps=None):
        res = self.denoise_branch(x)
        out_stats = res[0] if isinstance(res, tuple) else res
        z = self.reparameterize(out_stats, eps)
        return out_stats, self.detector(z)

    def forward_pair(self, x, x2, eps=None, eps2=None):
        """Two forward passes (the training step's original and flipped batch, denoiser_v2.py:295-311)
        as ONE pass over the stacked batch — twice the work per kernel launch, half the launches.  The
        blind-spot U-Net is stateless.  The detector's BatchNorm layers are told that the batch holds two
        passes: each half gets its own batch statistics and the running averages are updated half after
        half, which is what two separate calls do (no BatchNorm output depends on another layer's running
        average, so the final buffers are the same)."""
        B = x.shape[0]
        res = self.denoise_branch(torch.cat((x, x2), dim=0))
        both = res[0] if isinstance(res, tuple) else res
        out1, out2 = both[:B], both[B:]
        if not self.training:
            return (out1, self.detector(self.reparameterize(out1, eps))), \
                   (out2, self.detector(self.reparameterize(out2, eps2)))
        if eps is None or eps2 is None:
            z = torch.cat((self.reparameterize(out1, eps), self.reparameterize(out2, eps2)), dim=0)
        else:
            z = self.reparameterize(both, torch.cat((eps, eps2), dim=0))     # the same per-pixel arithmetic, one launch
        bns = [m for m in self.detector.modules() if isinstance(m, BatchNorm2d)]
        for m in bns:
            m.groups, m.count_outside = 2, True
        try:
            det = self.detector(z)
            torch._foreach_add_([m.num_batches_tracked for m in bns], 2)   # one launch instead of one per layer
        finally:
            for m in bns:
                m.groups, m.count_outside = 1, False
        return (out1, det[:B]), (out2, det[B:])
