"""Synthetic micrographs and patch batches of BASELINE.json's shape: white noise
+ Gaussian blobs, min-max normalised and quantised to uint8 like the reference's loader
(utils/loader.py:57-59); labelled centres carry CenterNet-style gaussian targets, everything
else is unlabelled (-1), sampled with 10 % positives like StratifiedCoordinateSampler
(datasets/sampler.py).  ``write_dataset`` puts such a set on disk in the formats the `joint` CLI reads
(MRC micrographs + the image / coordinate tables of README.md:29-47), with the labelled subset the README
describes (part of the particles inside a ~300x300 sub-region) and, beside it, the table of ALL planted
centres that recall / precision of the picks are measured against.
Used by bench.py, full_pipeline.py and the tests; data only, no kernels."""
import os

import numpy as np
import torch


def micrograph(idx, size=1024, blobs=None, seed=1234):
    rng = np.random.default_rng(seed + idx)
    img = rng.standard_normal((size, size), dtype=np.float32)
    nb = blobs if blobs is not None else int(200 * (size / 1024.0) ** 2)
    centres = rng.integers(72, size - 72, size=(nb, 2))
    yy, xx = np.mgrid[-16:17, -16:17]
    stamp = (-1.5 * np.exp(-(yy ** 2 + xx ** 2) / (2 * 4.0 ** 2))).astype(np.float32)
    for cy, cx in centres:
        img[cy - 16:cy + 17, cx - 16:cx + 17] += stamp
    lo, hi = img.min(), img.max()
    q = np.clip((img - lo) / (hi - lo) * 255.0, 0, 255).astype(np.uint8)
    in_box = (centres[:, 0] < 372) & (centres[:, 1] < 372) & (rng.random(nb) < 0.6)
    return q, centres, centres[in_box]


def patch_batches(n_batches, batch, micrographs, patch=64, seed=0, device="cuda"):
    """Pre-extracted training batches: list of (inp [B,1,P,P] float32 on device, target [B,1] host)."""
    rng = np.random.default_rng(seed)
    out = []
    half = patch // 2
    mics = [torch.from_numpy(m[0].astype(np.float32) / 255.0) for m in micrographs]
    for _ in range(n_batches):
        inp = torch.empty(batch, 1, patch, patch)
        tgt = torch.full((batch, 1), -1.0)
        for b in range(batch):
            m = int(rng.integers(0, len(micrographs)))
            size = mics[m].shape[0]
            labelled = micrographs[m][2]
            if rng.random() < 0.1 and len(labelled):
                cy, cx = labelled[int(rng.integers(0, len(labelled)))]
                dy, dx = rng.integers(-2, 3, size=2)
                y, x = int(cy + dy), int(cx + dx)
                tgt[b, 0] = float(np.exp(-(dy * dy + dx * dx) / (2 * 2.0 ** 2)))
            else:
                y, x = (int(v) for v in rng.integers(73, size - 140, size=2))
            inp[b, 0] = mics[m][y - half:y + half, x - half:x + half]
        out.append((inp.to(device), tgt))
    return out


def micrograph_standardised(idx, size=1024, blobs=None, seed=1234, sigma=0.25):
    """The same micrograph as ``micrograph`` (same noise, same centres, same labelled subset) before its min-max
    quantisation, as a STANDARDISED image: zero-mean noise of standard deviation ``sigma`` with particles 1.5 sigma
    deep — what a normalising preprocessor hands to ``save_png`` (utils/image.py:303-306: 8 bits over [-3, 3]).
    -> (uint8 PNG levels, centres, labelled)."""
    rng = np.random.default_rng(seed + idx)
    img = rng.standard_normal((size, size), dtype=np.float32)
    nb = blobs if blobs is not None else int(200 * (size / 1024.0) ** 2)
    centres = rng.integers(72, size - 72, size=(nb, 2))
    yy, xx = np.mgrid[-16:17, -16:17]
    stamp = (-1.5 * np.exp(-(yy ** 2 + xx ** 2) / (2 * 4.0 ** 2))).astype(np.float32)
    for cy, cx in centres:
        img[cy - 16:cy + 17, cx - 16:cx + 17] += stamp
    q = np.clip(np.rint((img * np.float32(sigma) + 3.0) * (255.0 / 6.0)), 0, 255).astype(np.uint8)
    in_box = (centres[:, 0] < 372) & (centres[:, 1] < 372) & (rng.random(nb) < 0.6)
    return q, centres, centres[in_box]


def write_dataset(root, n, size=1024, seed=1234, blobs=None, prefix="mic", fmt="mrc"):
    """n synthetic micrographs under `root` + ``images.txt`` (image_name, path), ``labels.txt`` (image_name, x_coord,
    y_coord: the labelled subset: 60 % of the centres inside the 300x300 corner window) and
    ``truth.txt`` (same columns, every planted centre).  x_coord = column, y_coord = row of the stored array.
    fmt "mrc": float32 MRC files — the loader min-max-scales them to [0, 1] (utils/loader.py:49-59), background level
    ~0.5; fmt "png": 8-bit PNG files of the standardised image (``micrograph_standardised``) — the loader maps them back
    to [-3, 3] (utils/loader.py:72-82), background level 0.  -> dict of the three paths + counts."""
    from . import micrograph_io, outputs
    if fmt not in ("mrc", "png"):
        raise ValueError("write_dataset: fmt is 'mrc' or 'png'")
    os.makedirs(root, exist_ok=True)
    images, labels, truth = ["image_name\tpath"], ["image_name\tx_coord\ty_coord"], ["image_name\tx_coord\ty_coord"]
    n_lab = n_all = 0
    for k in range(n):
        name = "%s%04d" % (prefix, k)
        path = os.path.join(root, name + "." + fmt)
        if fmt == "png":
            q, centres, labelled = micrograph_standardised(k, size=size, blobs=blobs, seed=seed)
            with open(path, "wb") as f:
                f.write(outputs.png_bytes(q))
        else:
            q, centres, labelled = micrograph(k, size=size, blobs=blobs, seed=seed)
            with open(path, "wb") as f:
                micrograph_io.write_mrc(f, q.astype(np.float32))
        images.append("%s\t%s" % (name, path))
        labels += ["%s\t%d\t%d" % (name, cx, cy) for cy, cx in labelled]
        truth += ["%s\t%d\t%d" % (name, cx, cy) for cy, cx in centres]
        n_lab += len(labelled)
        n_all += len(centres)
    out = {"images": os.path.join(root, "images.txt"), "labels": os.path.join(root, "labels.txt"),
           "truth": os.path.join(root, "truth.txt"), "n": n, "size": size, "labelled": n_lab, "planted": n_all,
           "format": fmt}
    for key, lines in (("images", images), ("labels", labels), ("truth", truth)):
        with open(out[key], "w") as f:
            f.write("\n".join(lines) + "\n")
    return out
