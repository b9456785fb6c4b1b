"""Particle-coordinate tables -> per-micrograph label rasters, restating
utils/coordinates.py of the reference:

* ``as_mask`` (:68-85): 1 inside the closed disc of `radius` around every particle, else 0 — the
  array whose non-zero pixels the stratified sampler treats as positives (radius 3, train.py:1057).
* ``as_gaussian`` (:87-97 with gaussian_radius :6-27, gaussian2D :30-38, draw_umich_gaussian
  :40-57): -1 background ("unlabelled") with a CenterNet gaussian window of radius
  int(gaussian_radius((bb, bb))) max-merged at every particle; the value at the sampled pixel is
  the PU-loss target of that patch.
* ``read_coordinates`` (utils/files.py:134-171): the default tab-separated table
  (image_name, x_coord, y_coord[, source, score]), RELION .star and EMAN .box; VIA .csv is not built.
* ``match_coordinates_to_images`` (:99-154): group per source, rasterise per image.

The reference also rasterises a half-resolution heat map (hm_small); nothing in the joint pipeline
reads it (denoiser_v2.py:263-272 only moves it to the device), so it is not produced here."""
import os

import numpy as np


def gaussian_radius(det_size, min_overlap=0.7):
    """Smallest of the three CornerNet radii for a box of det_size = (height, width)."""
    h, w = det_size
    roots = []
    for a, b, c in ((1.0, float(h + w), w * h * (1 - min_overlap) / (1 + min_overlap)),
                    (4.0, 2.0 * (h + w), (1 - min_overlap) * w * h),
                    (4.0 * min_overlap, -2.0 * min_overlap * (h + w), (min_overlap - 1) * w * h)):
        roots.append((b + np.sqrt(b * b - 4 * a * c)) / 2)   # the reference divides by 2, not 2a
    return min(roots)


def gaussian2d(shape, sigma=1.0):
    m, n = [(s - 1.0) / 2.0 for s in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def stamp_gaussian(heatmap, centre, radius, k=1):
    """Max-merge a (2r+1)^2 gaussian (sigma = diameter/6) clipped to the array, in place."""
    diameter = 2 * radius + 1
    g = gaussian2d((diameter, diameter), sigma=diameter / 6)
    x, y = int(centre[0]), int(centre[1])
    height, width = heatmap.shape[:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    dst = heatmap[y - top:y + bottom, x - left:x + right]
    src = g[radius - top:radius + bottom, radius - left:radius + right]
    if min(src.shape) > 0 and min(dst.shape) > 0:
        np.maximum(dst, src * k, out=dst)
    return heatmap


def as_gaussian(shape, x_coord, y_coord, bb=36):
    hm = np.full(shape, -1, dtype=np.float32)
    radius = max(0, int(gaussian_radius((bb, bb))))
    for x, y in zip(x_coord, y_coord):
        stamp_gaussian(hm, np.array([x, y]).astype(np.int32), radius)
    return hm


def as_mask(shape, x_coord, y_coord, radii):
    """uint8 [rows, cols]: 1 where (col-x)^2 + (row-y)^2 <= radius^2 for some particle.  Stamped per
    particle in its bounding window instead of over the whole grid (same pixels set)."""
    rows, cols = shape
    mask = np.zeros(shape, dtype=np.uint8)
    for x, y, r in zip(x_coord, y_coord, radii):
        x, y, r = int(x), int(y), int(r)
        y0, y1 = max(0, y - r), min(rows, y + r + 1)
        x0, x1 = max(0, x - r), min(cols, x + r + 1)
        if y0 >= y1 or x0 >= x1:
            continue
        yy, xx = np.ogrid[y0:y1, x0:x1]
        mask[y0:y1, x0:x1] |= ((xx - x) ** 2 + (yy - y) ** 2 <= r * r).astype(np.uint8)
    return mask


STAR_COLUMNS = {"AutopickFigureOfMerit": "score", "MicrographName": "image_name", "CoordinateX": "x_coord",
                "CoordinateY": "y_coord", "Voltage": "voltage", "DetectorPixelSize": "detector_pixel_size",
                "Magnification": "magnification", "AmplitudeContrast": "amplitude_contrast"}
_STAR_FLOATS = ("AutopickFigureOfMerit", "Voltage", "DetectorPixelSize", "Magnification", "AmplitudeContrast")


def parse_star(lines):
    """First ``data_`` block's ``loop_`` table (utils/star.py:18-100): column names lose the ``_rln``
    prefix and any trailing ``#k``; coordinates become int(float(.)), the known numeric columns float;
    the old ``ParticleScore`` column is renamed to ``AutopickFigureOfMerit``."""
    import pandas as pd
    it = iter(range(len(lines)))
    start = next((i for i in it if lines[i].startswith("data_")), None)
    if start is None:
        return None
    lines = lines[start + 1:]
    loop = next((i for i, l in enumerate(lines) if l.startswith("loop_")), None)
    if loop is not None:
        lines = lines[loop + 1:]
    columns, k = [], 0
    for k, raw in enumerate(lines):
        line = raw.strip()
        if not line.startswith("_"):
            break
        name = line[1:]
        if name.find("#") >= 0:
            name = name[:name.find("#")]
        if name.startswith("rln"):
            name = name[3:]
        columns.append(name.strip())
    content = []
    for raw in lines[k:]:
        line = raw.strip()
        if line.startswith("data"):
            break
        if line.startswith("#") or line.startswith(";") or line == "":
            continue
        content.append(line.split())
    table = pd.DataFrame(content, columns=columns)
    if "ParticleScore" in table.columns and "AutopickFigureOfMerit" not in table.columns:
        table["AutopickFigureOfMerit"] = table["ParticleScore"]
        table = table.drop("ParticleScore", axis=1)
    for c in ("CoordinateX", "CoordinateY"):
        if c in table:
            table[c] = table[c].astype(float).astype(int)
    for c in _STAR_FLOATS:
        if c in table:
            table[c] = table[c].astype(float)
    return table


def boxes_to_coordinates(boxes, image_name):
    """EMAN .box rows (x, y of the lower-left corner, width, height) -> centre coordinates
    (utils/conversions.py:13-44, no y inversion)."""
    import pandas as pd
    if len(boxes) < 1:
        return pd.DataFrame(columns=["x_coord", "y_coord", "image_name"])
    boxes = np.asarray(boxes)
    xy = np.stack([boxes[:, 0] + boxes[:, 2] // 2, boxes[:, 1] + boxes[:, 3] // 2], axis=1)
    table = pd.DataFrame(xy, columns=["x_coord", "y_coord"])
    table.insert(0, "image_name", [image_name] * len(table))
    return table


def read_coordinates(path):
    """-> pandas table with image_name, x_coord, y_coord (+ source / score when present)
    (utils/files.py:134-171): .txt/.tab tab-separated table, RELION .star, EMAN .box."""
    import pandas as pd
    ext = os.path.splitext(path)[1]
    if ext == ".star":
        with open(path) as f:
            table = parse_star(f.readlines())
        for star_name, ours in STAR_COLUMNS.items():
            if star_name in table.columns:
                table[ours] = table[star_name]
                table = table.drop(star_name, axis=1)
        table["image_name"] = table["image_name"].apply(lambda n: os.path.splitext(n)[0])
        return table
    if ext == ".box":
        rows = [[int(t) for t in line.split()[:4]] for line in open(path) if line.strip()]
        return boxes_to_coordinates(np.array(rows, dtype=int).reshape(-1, 4),
                                    os.path.basename(os.path.splitext(path)[0]))
    if ext in (".json", ".csv"):
        raise NotImplementedError("coordinate format %s (EMAN2 json / VIA csv) is not built" % ext)
    if ext not in (".txt", ".tab"):
        raise ValueError("Unknown coordinate file extension: %r" % ext)
    return pd.read_csv(path, sep="\t")


def coordinates_by_image(table):
    """{source: {image_name: int32 [n,2] (x, y)}}; tables without a source column use source 0."""
    out = {}
    if "source" in table:
        for (source, name), df in table.groupby(["source", "image_name"]):
            out.setdefault(source, {})[str(name)] = df[["x_coord", "y_coord"]].values.astype(np.int32)
    else:
        for name, df in table.groupby("image_name"):
            out.setdefault(0, {})[str(name)] = df[["x_coord", "y_coord"]].values.astype(np.int32)
    return out


def match_coordinates_to_images(table, images, radius=3, bb=24):
    """images: {source: {name: array [rows, cols]}} in load order ->
    {source: {name: (image, mask uint8, hm float32)}} with images that have no particles kept
    (empty mask, all -1 heat map)."""
    coords = coordinates_by_image(table) if table is not None else {}
    none = np.zeros((0, 2), dtype=np.int32)
    matched = {}
    for source, group in images.items():
        these = coords.get(source, {})
        for name, im in group.items():
            xy = these.get(name, none)
            shape = im.shape
            mask = as_mask(shape, xy[:, 0], xy[:, 1], [radius] * len(xy))
            hm = as_gaussian(shape, xy[:, 0], xy[:, 1], bb=bb)
            matched.setdefault(source, {})[name] = (im, mask, hm)
    return matched
