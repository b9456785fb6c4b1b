"""Drop-in for ``spr_pick.utils.algorithms.non_maximum_suppression`` (utils/algorithms.py:59-103)
running on the GPU (libsprk.so: sprk_nms2d)."""
import numpy as np
import torch

from . import _lib
from . import torch_ops  # noqa: F401  (registers torch.ops.sprk.nms2d)

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
            torch.ops.sprk.nms2d(score, int(r), thr, out_s, out_xy, cnt, ROUNDS_PER_CALL, resume, ws)
            n, undecided = cnt.tolist()
            if undecided == 0:
                break
            resume = 1
        if n <= cap:
            return out_s[:n], out_xy[:n]
        cap = n  # list overflowed (only possible with a caller-supplied max_out): redo with room


UPDATE_EMPTY_CONTAM = False   # see non_maximum_suppression


def _disk_offsets(r):
    ii, jj = np.meshgrid(np.arange(-r, r + 1), np.arange(-r, r + 1))
    keep = (ii ** 2 + jj ** 2) <= r * r
    return ii[keep], jj[keep]


def non_maximum_suppression(x, r, contam=None, threshold=-np.inf):
    """Same signature and return convention as the reference: ``x`` is an [H,W] score map (NumPy
    array or tensor), returns (scores float32[n], coords int32[n,2]) NumPy arrays.

    ``contam`` is the reference's suppressed-index set (utils/algorithms.py:77): flat indices already in it are never
    emitted and suppress nothing, and the reference ADDS every index its picks suppress (clip-to-H/W quirk included).
    Both are reproduced: the pre-suppressed pixels are taken out of the device score map (a score of -inf is never
    reached before the threshold stops the walk, which is exactly "skipped"), and the picks' disks are added to the
    set afterwards.  The reference's only caller passes a fresh ``set()`` and drops it (train.py:564); adding
    ~1000 indices per pick to a Python set costs seconds per micrograph, so an EMPTY set is left untouched unless
    ``algorithms.UPDATE_EMPTY_CONTAM`` is set."""
    if isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    else:
        t = x.detach().to(device="cuda", dtype=torch.float32)
    H, W = t.shape
    seeded = contam is not None and len(contam) > 0
    if seeded:
        idx = np.fromiter((i for i in contam if 0 <= i < H * W), dtype=np.int64)
        t = t.clone()
        t.view(-1)[torch.from_numpy(idx).to(t.device)] = float("-inf")
    s, c = nms_device(t, r, threshold)
    s, c = s.cpu().numpy(), c.cpu().numpy()
    if contam is not None and (seeded or UPDATE_EMPTY_CONTAM) and len(s):
        di, dj = _disk_offsets(int(r))
        yc = np.clip(c[:, 1:2].astype(np.int64) + di[None, :], 0, H)      # clip bounds H / W as in the reference
        xc = np.clip(c[:, 0:1].astype(np.int64) + dj[None, :], 0, W)
        contam.update(np.unique(yc * W + xc).tolist())
    return s, c
