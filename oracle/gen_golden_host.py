"""Golden vectors for the host-side rows next to the hot path (sampler, label
rasterisers, MRC parsing) — TEST INFRASTRUCTURE ONLY.

Runs the reference's own numpy-only modules (datasets/sampler.py, utils/coordinates.py,
utils/mrc.py) from where they lie under /root/reference, through oracle/ref_shim.py, and writes
inputs + expected outputs to tests/golden/host.npz.  Run here only (the reference never travels):

    python oracle/gen_golden_host.py
"""
import importlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    from oracle import ref_shim
    ref_shim.load()
    coords = importlib.import_module("spr_pick.utils.coordinates")
    sampler = importlib.import_module("spr_pick.datasets.sampler")
    mrc = importlib.import_module("spr_pick.utils.mrc")
    d = {}

    # ---- label rasterisers --------------------------------------------------------------
    shape = (120, 100)
    xs = np.array([10, 50, 99, 0, 52, 97], dtype=np.int32)
    ys = np.array([12, 60, 119, 0, 61, 3], dtype=np.int32)
    d["ras_shape"] = np.array(shape)
    d["ras_x"], d["ras_y"] = xs, ys
    d["ras_mask_r3"] = coords.as_mask(shape, xs, ys, np.array([3] * len(xs), dtype=np.int32))
    d["ras_mask_r0"] = coords.as_mask(shape, xs, ys, np.array([0] * len(xs), dtype=np.int32))
    d["ras_hm_bb24"] = coords.as_gaussian(shape, xs, ys, bb=24)
    d["ras_hm_bb36"] = coords.as_gaussian(shape, xs, ys, bb=36)
    d["ras_hm_empty"] = coords.as_gaussian(shape, xs[:0], ys[:0], bb=24)
    d["gaussian_radius"] = np.array([coords.gaussian_radius((b, b)) for b in (12, 24, 32, 36)])
    d["gaussian2d_13"] = coords.gaussian2D((13, 13), sigma=13 / 6)

    # ---- sampler ------------------------------------------------------------------------
    # datasets/sampler.py:145 forms i*2**56 + j*2**32 + c with j, c numpy uint32 scalars: fine under
    # the NumPy 1.x value-based casting it was written for, an OverflowError under this image's
    # NumPy 2.2 (NEP 50).  The pool's __next__ is wrapped here (at run time, nothing is edited) to
    # hand back Python ints, which restores the 1.x arithmetic without touching the RNG stream.
    _orig_next = sampler.ShuffledSampler.__next__

    def _next_as_ints(self):
        image, coord = _orig_next(self)
        return int(image), int(coord)

    sampler.ShuffledSampler.__next__ = _next_as_ints
    sampler.ShuffledSampler.next = _next_as_ints
    rng = np.random.RandomState(3)
    groups = []
    for shapes in (((230, 250), (250, 230)), ((240, 240),)):
        group = []
        for (r, c) in shapes:
            n = 12   # half inside the (swapped) margin window 72 < row < c-140, 72 < col < r-140, half anywhere
            px = np.concatenate([rng.randint(74, r - 142, size=n // 2), rng.randint(0, c, size=n // 2)]).astype(np.int32)
            py = np.concatenate([rng.randint(74, c - 142, size=n // 2), rng.randint(0, r, size=n // 2)]).astype(np.int32)
            group.append(coords.as_mask((r, c), px, py, np.array([3] * n, dtype=np.int32)))
        groups.append(group)
    for g, group in enumerate(groups):
        for i, m in enumerate(group):
            d["smp_label_%d_%d" % (g, i)] = m
        P, U = sampler.enumerate_pu_coordinates(group)
        d["smp_P_%d" % g] = np.stack([P["image"], P["coord"]], 1).astype(np.int64)
        d["smp_U_%d" % g] = np.stack([U["image"], U["coord"]], 1).astype(np.int64)
    for balance, tag in ((0.1, "b10"), (None, "bnone")):
        s = sampler.StratifiedCoordinateSampler(groups, balance=balance, size=400,
                                                random=np.random.RandomState(7))
        d["smp_draws_" + tag] = np.array([next(s) for _ in range(400)], dtype=np.int64)
        d["smp_weights_" + tag] = s.weights
    s = sampler.StratifiedCoordinateSampler(groups, balance=0.1, random=np.random.RandomState(7))
    d["smp_default_size"] = np.array(len(s))

    # ---- MRC ----------------------------------------------------------------------------
    img = np.random.RandomState(5).randn(1, 20, 28).astype(np.float32)
    buf = io.BytesIO()
    mrc.write(buf, img)
    raw = buf.getvalue()
    arr, hdr, ext = mrc.parse(raw)
    d["mrc2_bytes"] = np.frombuffer(raw, dtype=np.uint8)
    d["mrc2_array"] = arr
    d["mrc2_header_nums"] = np.array([hdr.nx, hdr.ny, hdr.nz, hdr.mode, hdr.next, hdr.amin, hdr.amax, hdr.amean,
                                      hdr.rms], dtype=np.float64)
    for mode, dt in ((0, np.int8), (1, np.int16), (6, np.uint16)):
        data = (np.random.RandomState(mode).randint(0, 100, size=(1, 9, 11))).astype(dt)
        extb = bytes(range(16))
        h = hdr._replace(nx=11, ny=9, nz=1, mode=mode, next=len(extb))
        raw_m = mrc.header_struct.pack(*list(h)) + extb + data.tobytes() + b"\x00" * 6   # trailing junk is clipped
        a, hh, e = mrc.parse(raw_m)
        assert e == extb
        d["mrc%d_bytes" % mode] = np.frombuffer(raw_m, dtype=np.uint8)
        d["mrc%d_array" % mode] = a
    stack = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    buf = io.BytesIO()
    mrc.write(buf, stack)
    d["mrc_stack_bytes"] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
    d["mrc_stack_array"] = mrc.parse(buf.getvalue())[0]

    # ---- STAR / box tables ----------------------------------------------------------------
    import pandas as pd
    import tempfile
    files = importlib.import_module("spr_pick.utils.files")
    star = importlib.import_module("spr_pick.utils.star")
    conv = importlib.import_module("spr_pick.utils.conversions")
    star_text = ("# version 30001\n\ndata_\n\nloop_\n_rlnCoordinateX #1\n_rlnCoordinateY #2\n_rlnMicrographName #3\n"
                 "_rlnParticleScore #4\n_rlnVoltage #5\n# a comment\n10.7\t20.2\tmicA.mrc\t0.5\t300\n\n"
                 "33\t44.9\tsub/micB.tiff\t-1.25\t300\ndata_next\n1\t2\tmicC.mrc\t9\t1\n")
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "p.star"), "w") as f:
        f.write(star_text)
    t = files.read_coordinates(os.path.join(tmp, "p.star"))
    d["star_text"] = np.frombuffer(star_text.encode(), dtype=np.uint8)
    d["star_columns"] = np.array(list(t.columns))
    d["star_xy"] = t[["x_coord", "y_coord"]].values.astype(np.int64)
    d["star_score"] = t["score"].values.astype(np.float64)
    d["star_names"] = np.array(list(t["image_name"]))
    with open(os.path.join(tmp, "micZ.box"), "w") as f:
        f.write("10 20 30 40\n  5   6   7   9 extra\n")
    b = files.read_coordinates(os.path.join(tmp, "micZ.box"))
    d["box_xy"] = b[["x_coord", "y_coord"]].values.astype(np.int64)
    d["box_names"] = np.array(list(b["image_name"]))
    table = pd.DataFrame({"image_name": ["m1", "m2"], "x_coord": [3, 4], "y_coord": [5, 6], "score": [0.25, 1.5]})
    out = io.StringIO()
    star.write(conv.coordinates_to_star(table, image_ext=".mrc"), out)
    d["star_written"] = np.frombuffer(out.getvalue().encode(), dtype=np.uint8)

    np.savez_compressed(os.path.join(OUT, "host.npz"), **d)
    print("wrote", os.path.join(OUT, "host.npz"), "with", len(d), "arrays")


if __name__ == "__main__":
    main()
