"""CPU restatement of the reference's greedy 2-D non-maximum suppression.

TEST INFRASTRUCTURE ONLY (see oracle/networks.py header).

Follows /root/reference/spr_pick/utils/algorithms.py:59-103 as called at
spr_pick/train.py:564 (contam = empty set, threshold = 0.02):

  * visit pixels by descending score, stop at the first score <= threshold;
  * an unsuppressed pixel i is emitted as (score, xx = i % W, yy = i // W) and every
    flat index  clip(yy+di, 0, H) * W + clip(xx+dj, 0, W)  with di^2+dj^2 <= r^2 is
    added to the suppressed set.  The clip bounds are H and W (not H-1, W-1), so an
    x overflow lands on column 0 of the NEXT row and a y overflow lands past the
    array — both reproduced here exactly.

The one thing the reference leaves implementation-defined is the order of equal
scores (``np.argsort`` default kind is not stable).  This restatement — and the
HIP kernel — define it as: score descending, then flat index descending
(= a stable ascending argsort, reversed).

``nms_literal`` is the plain-Python statement (small inputs);
``nms_c`` calls the same algorithm compiled from oracle/nms_c.c (large inputs,
bench.py's cpu_baseline leg).
"""
import ctypes
import os

import numpy as np


def disk_offsets(r):
    return [(di, dj) for di in range(-r, r + 1) for dj in range(-r, r + 1)
            if di * di + dj * dj <= r * r]


def nms_literal(x, r, threshold=-np.inf, contam=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    H, W = x.shape
    A = x.ravel()
    order = np.argsort(A, kind="stable")[::-1]
    offs = disk_offsets(r)
    # the reference walks and MUTATES the caller's set (utils/algorithms.py:77,98-101): indices already in it are
    # neither emitted nor do they suppress anything
    suppressed = contam if contam is not None else set()
    scores, coords = [], []
    for i in order:
        i = int(i)
        if A[i] <= threshold:
            break
        if i in suppressed:
            continue
        xx, yy = i % W, i // W
        scores.append(A[i])
        coords.append((xx, yy))
        for di, dj in offs:
            yc = min(max(yy + di, 0), H)
            xc = min(max(xx + dj, 0), W)
            suppressed.add(yc * W + xc)
    return (np.asarray(scores, dtype=np.float32),
            np.asarray(coords, dtype=np.int32).reshape(-1, 2))


_lib = None


def _load_c():
    global _lib
    if _lib is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "_build", "liboracle_nms.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle C library not built: run `make -C oracle` "
                               "(or __graft_entry__.build())")
        _lib = ctypes.CDLL(path)
        _lib.oracle_nms2d.restype = ctypes.c_long
        _lib.oracle_nms2d.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    return _lib


def nms_c(x, r, threshold=-np.inf):
    lib = _load_c()
    x = np.ascontiguousarray(x, dtype=np.float32)
    H, W = x.shape
    cap = H * W
    scores = np.empty(cap, dtype=np.float32)
    coords = np.empty((cap, 2), dtype=np.int32)
    thr = np.float32(max(threshold, -3.0e38))
    n = lib.oracle_nms2d(x.ctypes.data, H, W, int(r), ctypes.c_float(thr),
                         scores.ctypes.data, coords.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("oracle_nms2d failed: %d" % n)
    return scores[:n].copy(), coords[:n].copy()
