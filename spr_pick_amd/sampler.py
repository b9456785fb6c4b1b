"""Patch-centre sampling of the joint trainer, restating datasets/sampler.py:

* ``enumerate_pu_coordinates`` (:14-55): per label image [rows r, cols c], the candidate centres are
  the flat indices whose (row, col) satisfy 72 < row < c-140 and 72 < col < r-140 (the reference
  compares rows with the column count and vice versa — kept); U = all candidates in raster order,
  P = those with a non-zero label.  The reference walks every pixel in a Python loop; here it is a
  NumPy mask — same arrays, same order.
* ``ShuffledSampler`` (:57-79): endless pass over an array, reshuffled in place with
  ``random.shuffle`` each time it is exhausted (and before the first draw).
* ``StratifiedCoordinateSampler`` (:81-154): alternates between the P and U pools of each group so
  that the running draw frequencies track `weights` (balance = 0.1 -> 10 % positives); a draw is
  returned as the hash  group*2^56 + image*2^32 + coord  (``decode_index`` undoes it).

Given the same ``numpy.random.RandomState`` the streams are identical to the reference's: the RNG
is consumed by the same calls in the same order (shuffle over arrays of the same length, then
``choice(len(weights), p=weights)`` per draw).  Pools are kept as packed uint64 keys
(image << 32 | coord) instead of a structured (image, coord) array; legacy ``shuffle`` draws
depend only on the length."""
import numpy as np

ROW_LO = COL_LO = 72
MARGIN_HI = 140


def enumerate_pu_coordinates(labels):
    """labels: list of 2-D arrays -> (P, U) uint64 arrays of image << 32 | coord."""
    p_parts, u_parts = [], []
    for image, y in enumerate(labels):
        y = np.asarray(y)
        r, c = y.shape
        rows = np.arange(r)[:, None]
        cols = np.arange(c)[None, :]
        ok = (rows > ROW_LO) & (rows < c - MARGIN_HI) & (cols > COL_LO) & (cols < r - MARGIN_HI)
        coord = np.flatnonzero(ok).astype(np.uint64)
        key = (np.uint64(image) << np.uint64(32)) | coord
        u_parts.append(key)
        p_parts.append(key[y.ravel()[coord.astype(np.int64)] != 0])
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)  # noqa: E731
    return cat(p_parts), cat(u_parts)


class ShuffledSampler:
    def __init__(self, x, random=np.random):
        self.x = x
        self.random = random
        self.i = len(self.x)

    def __len__(self):
        return len(self.x)

    def __next__(self):
        if self.i >= len(self.x):
            self.random.shuffle(self.x)
            self.i = 0
        sample = self.x[self.i]
        self.i += 1
        return sample

    def __iter__(self):
        return self


class StratifiedCoordinateSampler:
    def __init__(self, labels, balance=0.5, size=None, random=np.random):
        n = len(labels)
        self.groups = []
        self.weights = np.zeros(n * 2)
        self.proportions = np.zeros((n, 2))
        for g, group in enumerate(labels):
            P, U = enumerate_pu_coordinates(group)
            if len(U) == 0:
                raise ValueError("no patch centre satisfies the 72 / 140 px margin rule in group %d "
                                 "(micrographs must be larger than 213 px)" % g)
            self.groups += [ShuffledSampler(P, random=random), ShuffledSampler(U, random=random)]
            self.proportions[g, 0] = (len(U) - len(P)) / len(U)
            self.proportions[g, 1] = len(P) / len(U)
            p = self.proportions[g, 1] if balance is None else balance
            self.weights[2 * g] = p / n
            self.weights[2 * g + 1] = (1 - p) / n
        if size is None:
            sizes = np.array([len(s) for s in self.groups])
            size = int(np.round(np.min(sizes / self.weights)))
        self.size = size
        self.history = np.zeros_like(self.weights)
        self.random = random

    def __len__(self):
        return self.size

    def __next__(self):
        n = self.history.sum()
        weights = self.weights
        if n > 0:
            weights = weights - self.history / n
            weights[weights < 0] = 0
            n = weights.sum()
            weights = weights / n if n > 0 else np.ones_like(weights) / len(weights)
        i = self.random.choice(len(weights), p=weights)
        self.history[i] += 1
        if np.all(self.history / self.history.sum() == self.weights):
            self.history[:] = 0
        pool = self.groups[i]
        if len(pool) == 0:
            raise ValueError("the sampler drew from an empty pool (no labelled particle inside the margins)")
        key = int(next(pool))
        return (i // 2) * 2 ** 56 + (key >> 32) * 2 ** 32 + (key & 0xFFFFFFFF)

    def __iter__(self):
        for _ in range(self.size):
            yield next(self)


def decode_index(h):
    """hash -> (group, image, coord)   (datasets/micrograph.py:62-67)."""
    g = h // 2 ** 56
    h -= g * 2 ** 56
    i = h // 2 ** 32
    return g, i, h - i * 2 ** 32


def sequential_indices(n_items, num_samples=None):
    """FixedLengthSampler(shuffled=False) (datasets/sampler.py:157-207): 0..n-1, wrapping."""
    total = n_items if num_samples is None else num_samples
    return [k % n_items for k in range(total)]
