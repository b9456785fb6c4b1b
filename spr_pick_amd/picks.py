"""Pick post-filter and ``*_scores.txt`` writer — the consumer of the NMS output in the reference
(train.py:557-571): keep picks farther than 30 px from every border of the un-padded score map
and write ``name<TAB>coords[i,1]<TAB>coords[i,0]<TAB>score`` under the header
``image_name x_coord y_coord score``.  (The reference's tensors are H/W-transposed, so array
row = real x: column order (row, col) is what lands in the x/y columns — reproduced as is.)"""
import numpy as np

BORDER = 30


def filter_picks(scores, coords, shape, border=BORDER):
    """coords[:,0] = column (xx), coords[:,1] = row (yy) as returned by non_maximum_suppression;
    shape = (rows, cols) of the score map."""
    scores = np.asarray(scores)
    coords = np.asarray(coords).reshape(-1, 2)
    x_max, y_max = shape[0] - border, shape[1] - border
    keep = (coords[:, 1] > border) & (coords[:, 1] < x_max) & (coords[:, 0] > border) & (coords[:, 0] < y_max)
    return scores[keep], coords[keep]


def write_scores(path, name, scores, coords, shape, border=BORDER):
    s, c = filter_picks(scores, coords, shape, border)
    with open(path, "w") as f:
        print("image_name\tx_coord\ty_coord\tscore", file=f)
        for i in range(len(s)):
            print(name + "\t" + str(c[i, 1]) + "\t" + str(c[i, 0]) + "\t" + str(s[i]), file=f)
    return len(s)


def read_scores(path):
    """``*_scores.txt`` -> (names list, xy int64 [n,2] = (x_coord, y_coord), scores float64 [n])."""
    names, xy, scores = [], [], []
    with open(path) as f:
        head = f.readline().rstrip("\n").split("\t")
        if head != ["image_name", "x_coord", "y_coord", "score"]:
            raise ValueError("%s: not a scores table (header %r)" % (path, head))
        for line in f:
            n, x, y, s = line.rstrip("\n").split("\t")
            names.append(n); xy.append((int(x), int(y))); scores.append(float(s))
    return names, np.asarray(xy, dtype=np.int64).reshape(-1, 2), np.asarray(scores, dtype=np.float64)


def match_picks(pred_xy, pred_scores, truth_xy, radius):
    """One-to-one assignment of picks to true centres: picks in descending score order, each takes the nearest
    still-unmatched centre within `radius` (Euclidean, closed) or counts as a false positive.
    -> (order [n] = pick indices by descending score (ties: lower index first), hit bool [n] in that order)."""
    pred_xy = np.asarray(pred_xy, dtype=np.float64).reshape(-1, 2)
    truth_xy = np.asarray(truth_xy, dtype=np.float64).reshape(-1, 2)
    order = np.argsort(-np.asarray(pred_scores, dtype=np.float64), kind="stable")
    free = np.ones(len(truth_xy), dtype=bool)
    hit = np.zeros(len(order), dtype=bool)
    r2 = float(radius) ** 2
    for k, i in enumerate(order):
        if not free.any():
            break
        d2 = ((truth_xy - pred_xy[i]) ** 2).sum(axis=1)
        d2[~free] = np.inf
        j = int(np.argmin(d2))
        if d2[j] <= r2:
            free[j] = False
            hit[k] = True
    return order, hit


def detection_metrics(per_image, radius, thresholds=(0.5,)):
    """per_image: iterable of (pred_xy, pred_scores, truth_xy).  Pools the matched picks of all images ->
    {"n_truth", "n_picks", "average_precision", "best_f1": {...}, "at": {thr: {precision, recall, picks}}}."""
    scores, hits, n_truth = [], [], 0
    for xy, s, truth in per_image:
        order, hit = match_picks(xy, s, truth, radius)
        scores.append(np.asarray(s, dtype=np.float64)[order]); hits.append(hit)
        n_truth += len(np.asarray(truth).reshape(-1, 2))
    scores = np.concatenate(scores) if scores else np.zeros(0)
    hits = np.concatenate(hits) if hits else np.zeros(0, dtype=bool)
    o = np.argsort(-scores, kind="stable")
    scores, hits = scores[o], hits[o]
    tp = np.cumsum(hits)
    k = np.arange(1, len(hits) + 1)
    prec = tp / np.maximum(k, 1)
    rec = tp / max(n_truth, 1)
    ap = float(np.sum(prec[hits]) / max(n_truth, 1)) if len(hits) else 0.0
    out = {"n_truth": int(n_truth), "n_picks": int(len(hits)), "radius": float(radius), "average_precision": ap, "at": {}}
    if len(hits):
        f1 = 2 * prec * rec / np.maximum(prec + rec, 1e-30)
        b = int(np.argmax(f1))
        out["best_f1"] = {"f1": float(f1[b]), "precision": float(prec[b]), "recall": float(rec[b]),
                          "threshold": float(scores[b]), "picks": b + 1}
    for thr in thresholds:
        m = int((scores > thr).sum())
        t = int(tp[m - 1]) if m else 0
        out["at"][float(thr)] = {"precision": t / m if m else 0.0, "recall": t / max(n_truth, 1), "picks": m}
    return out
