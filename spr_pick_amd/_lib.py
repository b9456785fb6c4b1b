"""ctypes binding of libsprk.so (include/sprk.h).  The product path has no fallback: if the
library is missing or a call fails, this module raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPRK_LIB") or os.path.join(_HERE, "libsprk.so")   # SPRK_LIB: kernel A/B experiments

ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2
DT_F32, DT_BF16, DT_F16 = 0, 1, 2          # SPRK_DT_*: precision of the MFMA operands (include/sprk.h)
DT_NAIVE = 0x200                           # SPRK_DT_NAIVE: plain per-output-element kernels for this call (cross-check)
DT_WPREP = 0x800                           # SPRK_DT_WPREP: the workspace already holds the transformed weights
DT_X16, DT_Y16 = 0x8000, 0x10000           # SPRK_DT_X16 / _Y16: the activation inputs / the output of a call are 16-bit tensors
DT_PIN = 0x400                             # SPRK_DT_PIN: kernel choice by layer structure only (inference: tiled == whole)
DT_FORCE = 0x100                           # SPRK_DT_FORCE: 16-bit kernel wherever it exists (tests), not only where faster
DTYPES = {"f32": DT_F32, "fp32": DT_F32, "bf16": DT_BF16, "f16": DT_F16, "fp16": DT_F16,
          "bf16!": DT_BF16 | DT_FORCE, "f16!": DT_F16 | DT_FORCE}

c_f = ctypes.c_void_p      # device float*
c_i = ctypes.c_int
c_vp = ctypes.c_void_p
c_sz = ctypes.c_size_t


class ConvGeom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("N", "C1", "C2", "Hin", "Win", "up1", "Cout", "Hout", "Wout",
                 "KH", "KW", "stride", "dil", "pad_top", "pad_left", "dtype")]


class ConvEpilogue(ctypes.Structure):
    _fields_ = [("bias", c_vp), ("scale", c_vp), ("shift", c_vp), ("res", c_vp),
                ("res_h", ctypes.c_int32), ("res_w", ctypes.c_int32), ("res_off", ctypes.c_int32),
                ("act", ctypes.c_int32), ("up2", ctypes.c_int32)]


class AdamItem(ctypes.Structure):
    """sprk_adam_item (include/sprk.h)."""
    _fields_ = [("p", c_vp), ("g", c_vp), ("m", c_vp), ("v", c_vp), ("n", ctypes.c_long)]


class ReduceItem(ctypes.Structure):
    """sprk_reduce_item: a pending second-stage sum (include/sprk.h)."""
    _fields_ = [("src", c_vp), ("dst", c_vp), ("kind", ctypes.c_int32), ("parts", ctypes.c_int32), ("n", ctypes.c_int32),
                ("K", ctypes.c_int32), ("Cout", ctypes.c_int32), ("CoutP", ctypes.c_int32)]


class WprepItem(ctypes.Structure):
    """sprk_wprep_item: a weight transform as data (opaque; include/sprk.h)."""
    _fields_ = [("w", c_vp), ("dst", c_vp), ("kind", ctypes.c_int32), ("blocks", ctypes.c_int32), ("p", ctypes.c_int32 * 12)]


_SIGS = {
    "sprk_last_error": (ctypes.c_char_p, []),
    "sprk_version": (c_i, []),
    "sprk_struct_bytes": (c_sz, [c_i]),
    "sprk_launch_count": (ctypes.c_long, []),
    "sprk_wino_launch_count": (ctypes.c_long, []),
    "sprk_conv16_launch_count": (ctypes.c_long, []),
    "sprk_wgrad16_launch_count": (ctypes.c_long, []),
    "sprk_conv2d_fwd_ws_bytes": (c_sz, [ctypes.POINTER(ConvGeom)]),
    "sprk_conv2d_fwd": (c_i, [c_f, c_f, c_f, c_f, ctypes.POINTER(ConvGeom), ctypes.POINTER(ConvEpilogue), c_vp, c_sz, c_vp]),
    "sprk_conv2d_fwd_wprep": (c_i, [c_f, ctypes.POINTER(ConvGeom), ctypes.POINTER(ConvEpilogue), c_vp, c_sz, ctypes.POINTER(WprepItem)]),
    "sprk_conv2d_bwd_data_wprep": (c_i, [c_f, ctypes.POINTER(ConvGeom), c_vp, c_sz, ctypes.POINTER(WprepItem)]),
    "sprk_prepare_weights": (c_i, [ctypes.POINTER(WprepItem), c_i, c_vp]),
    "sprk_head1x1_fwd_ws_bytes": (c_sz, [c_i, c_i]),
    "sprk_head1x1_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, ctypes.c_long, c_vp, c_sz, c_vp]),
    "sprk_head1x1_unrot_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "sprk_conv2d_bwd_data_ws_bytes": (c_sz, [ctypes.POINTER(ConvGeom)]),
    "sprk_conv2d_bwd_data": (c_i, [c_f, c_f, c_f, ctypes.POINTER(ConvGeom), c_vp, c_sz, c_vp]),
    "sprk_conv2d_bwd_data_masked": (c_i, [c_f, c_f, c_f, ctypes.POINTER(ConvGeom), c_f, c_i, c_vp, c_sz, c_vp]),
    "sprk_conv2d_bwd_weight_ws_bytes": (c_sz, [ctypes.POINTER(ConvGeom)]),
    "sprk_conv2d_storage16": (c_i, [ctypes.POINTER(ConvGeom), ctypes.POINTER(ConvEpilogue)]),
    "sprk_conv2d_bwd_weight": (c_i, [c_f, c_f, c_f, c_f, ctypes.POINTER(ConvGeom), c_vp, c_sz, c_vp]),
    "sprk_conv2d_bwd_weight_partial": (c_i, [c_f, c_f, c_f, c_f, ctypes.POINTER(ConvGeom), c_vp, c_sz, ctypes.POINTER(ReduceItem), c_vp]),
    "sprk_act_bwd_partial": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, ctypes.c_long, c_i, c_vp, c_sz, ctypes.POINTER(ReduceItem), c_vp]),
    "sprk_reduce_items": (c_i, [ctypes.POINTER(ReduceItem), c_i, c_vp]),
    "sprk_act_bwd_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "sprk_act_bwd": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, ctypes.c_long, c_i, c_vp, c_sz, c_vp]),
    "sprk_concat_up_bwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "sprk_shift_maxpool2_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "sprk_shift_maxpool2_bwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "sprk_rot4_stack_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_vp]),
    "sprk_rot4_stack_bwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_vp]),
    "sprk_unrot4_shift_concat_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_vp]),
    "sprk_unrot4_shift_concat_bwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_vp]),
    "sprk_bn_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "sprk_bn_train_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, ctypes.c_float, ctypes.c_float, c_i, c_vp, c_sz, c_vp]),
    "sprk_bn_eval_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, ctypes.c_float, c_i, c_vp]),
    "sprk_bn_train_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "sprk_crop_add_fwd": (c_i, [c_f, c_f, c_f, ctypes.c_long, c_i, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "sprk_crop_embed_bwd": (c_i, [c_f, c_f, ctypes.c_long, c_i, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "sprk_noise_std_fwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_vp]),
    "sprk_noise_std_bwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_vp]),
    "sprk_joint_loss_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, ctypes.c_float, ctypes.c_float, c_vp]),
    "sprk_joint_loss_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, ctypes.c_float, ctypes.c_float, c_vp]),
    "sprk_reparam_fwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_vp]),
    "sprk_reparam_bwd": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_vp]),
    "sprk_adam_multi": (c_i, [c_vp, c_vp, c_i, c_i, c_f, c_f, c_f, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_vp]),
    "sprk_pu_loss": (c_i, [c_f, c_f, c_f, c_i, ctypes.c_float, c_f, c_f, c_vp]),
    "sprk_sigmoid_clamp_fwd": (c_i, [c_f, c_f, ctypes.c_long, c_vp]),
    "sprk_sigmoid_clamp_bwd": (c_i, [c_f, c_f, c_f, ctypes.c_long, c_vp]),
    "sprk_ssdn_ws_bytes": (c_sz, [c_i, c_i]),
    "sprk_ssdn_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "sprk_ssdn_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "sprk_nms2d_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "sprk_nms2d": (c_i, [c_f, c_i, c_i, c_i, ctypes.c_float, c_f, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_sz, c_vp]),
    "sprk_gather_patches": (c_i, [c_vp, c_i, c_vp, c_vp, c_vp, c_f, c_i, c_i, c_i, c_vp]),
    "sprk_prof_enable": (None, [c_i]),
    "sprk_prof_collect": (c_i, [c_i, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "sprk_prof_collect_bytes": (c_i, [c_i, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_double)]),
}

EXPORTS = tuple(_SIGS)
ABI_VERSION = 410          # SPRK_ABI_VERSION of the include/sprk.h these signatures were written against
_lib = None


class SprkError(RuntimeError):
    pass


def lib():
    """Load libsprk.so (once).  Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SprkError("libsprk.so not found at %s — build it with `make -C spr_pick_amd/csrc` "
                            "or __graft_entry__.build(); spr_pick_amd has no fallback path" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        # a stale library (or a newer one) would read past the geometry struct or misplace arguments silently
        if L.sprk_version() != ABI_VERSION:
            raise SprkError("libsprk.so at %s has ABI version %d, this binding needs %d — rebuild it "
                            "(make -C spr_pick_amd/csrc)" % (LIB_PATH, L.sprk_version(), ABI_VERSION))
        for which, st in enumerate((ConvGeom, ConvEpilogue, ReduceItem, AdamItem, WprepItem)):
            if L.sprk_struct_bytes(which) != ctypes.sizeof(st):
                raise SprkError("libsprk.so: sizeof(%s) is %d in the library, %d in the binding"
                                % (st.__name__, L.sprk_struct_bytes(which), ctypes.sizeof(st)))
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise SprkError("%s failed (%d): %s" % (what, rc, lib().sprk_last_error().decode()))
