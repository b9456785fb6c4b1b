"""Why two greedy-NMS pick lists differ when their score maps agree to delta — the explicit near-tie list.

Greedy NMS (utils/algorithms.py:59-103) is discontinuous in the scores: two candidates closer than r whose scores are
nearly equal swap when a rounding-level perturbation changes their order, a candidate within delta of the threshold
appears or disappears, and every such swap can cascade along a chain of neighbours.  But nothing else can happen.  With
A-only / B-only the picks made on one map only, link an A-only and a B-only pick when they are within r of each other.
Claim (proved by looking at the element of a connected component with the highest score): every component contains a
ROOT — an (A-only p, B-only q) pair within r whose score order differs between the maps, hence |A[p] - A[q]| <= 2 delta,
or a pick whose score is within delta of the threshold.  `explain` returns the roots and raises if some component has
none: then the two lists differ for a reason that rounding cannot explain."""
import numpy as np


def _sets(ca, cb):
    a = set(map(tuple, np.asarray(ca).tolist()))
    b = set(map(tuple, np.asarray(cb).tolist()))
    return a, b


def jaccard(ca, cb):
    a, b = _sets(ca, cb)
    return len(a & b) / max(len(a | b), 1)


def explain(map_a, map_b, coords_a, coords_b, r, threshold, delta=None):
    """coords: [n,2] (x = column, y = row).  -> dict(a_only, b_only, roots, components, delta)."""
    map_a, map_b = np.asarray(map_a, dtype=np.float64), np.asarray(map_b, dtype=np.float64)
    if delta is None:
        delta = float(np.abs(map_a - map_b).max())
    a, b = _sets(coords_a, coords_b)
    a_only, b_only = sorted(a - b), sorted(b - a)
    nodes = [("a", p) for p in a_only] + [("b", q) for q in b_only]
    parent = list(range(len(nodes)))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    roots = []
    root_nodes = set()
    for i, p in enumerate(a_only):
        for j, q in enumerate(b_only):
            if (p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 <= r * r:
                parent[find(i)] = find(len(a_only) + j)
                sa = map_a[p[1], p[0]] - map_a[q[1], q[0]]
                if abs(sa) <= 2 * delta:
                    roots.append(("near-tie", p, q, float(sa)))
                    root_nodes.update((i, len(a_only) + j))
    for i, (side, p) in enumerate(nodes):
        s = map_a[p[1], p[0]] if side == "a" else map_b[p[1], p[0]]
        if abs(s - threshold) <= delta:
            roots.append(("threshold", p, None, float(s - threshold)))
            root_nodes.add(i)
    comps = {}
    for i in range(len(nodes)):
        comps.setdefault(find(i), []).append(i)
    for members in comps.values():
        if not any(m in root_nodes for m in members):
            raise AssertionError("pick lists differ without a near-tie or threshold cause: %s (delta %.3e)"
                                 % ([nodes[m] for m in members], delta))
    return {"a_only": a_only, "b_only": b_only, "roots": roots, "components": len(comps), "delta": delta,
            "jaccard": len(a & b) / max(len(a | b), 1)}
