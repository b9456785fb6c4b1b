"""Drop-in for ``spr_pick.utils.algorithms.non_maximum_suppression`` (utils/algorithms.py:59-103)
running on the GPU (libsprk.so: sprk_nms2d)."""
import ctypes

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream

ROUNDS_PER_CALL = 12


def _max_picks(H, W, r):
    if r <= 0:
        return H * W
    # picks are pairwise more than r apart: disks of radius r/2 around them are disjoint
    bound = int((H + r) * (W + r) / (0.7853 * r * r)) + 16
    return max(1, min(H * W, bound))


def nms_device(score, r, threshold, max_out=None):
    """score: [H,W] float32 CUDA tensor.  Returns (scores[n], coords[n,2]) CUDA tensors, picks in
    descending-score order, coords[:,0] = x (column), coords[:,1] = y (row)."""
    L = _lib.lib()
    if not score.is_cuda or score.dtype != torch.float32 or score.dim() != 2:
        raise _lib.SprkError("nms_device expects a 2-D float32 CUDA tensor")
    score = score.contiguous()
    H, W = score.shape
    thr = float(max(threshold, -3.0e38))
    cap = int(max_out) if max_out else _max_picks(H, W, int(r))
    while True:
        out_s = torch.empty(cap, dtype=torch.float32, device=score.device)
        out_xy = torch.empty((cap, 2), dtype=torch.int32, device=score.device)
        cnt = torch.zeros(2, dtype=torch.int32, device=score.device)
        nb = L.sprk_nms2d_ws_bytes(H, W, cap)
        ws = torch.empty(nb, dtype=torch.uint8, device=score.device)
        resume = 0
        while True:
            _lib.check(L.sprk_nms2d(_p(score), H, W, int(r), ctypes.c_float(thr), _p(out_s), _p(out_xy), _p(cnt), cap,
                                    ROUNDS_PER_CALL, resume, _p(ws), nb, _stream(score)), "sprk_nms2d")
            n, undecided = cnt.tolist()
            if undecided == 0:
                break
            resume = 1
        if n <= cap:
            return out_s[:n], out_xy[:n]
        cap = n  # list overflowed (only possible with a caller-supplied max_out): redo with room


def non_maximum_suppression(x, r, contam=None, threshold=-np.inf):
    """Same signature and return convention as the reference: ``x`` is an [H,W] score map (NumPy
    array or tensor), returns (scores float32[n], coords int32[n,2]) NumPy arrays.  ``contam`` is the
    reference's pre-suppressed index set; its only call site passes an empty set (train.py:564)."""
    if contam:
        raise NotImplementedError("non-empty `contam` is not on the hot path (reference caller passes set())")
    if isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    else:
        t = x.detach().to(device="cuda", dtype=torch.float32)
    s, c = nms_device(t, r, threshold)
    return s.cpu().numpy(), c.cpu().numpy()
