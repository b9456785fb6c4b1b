"""Micrograph files -> arrays, the way the reference's loaders hand them to the network
(SURVEY.md §8f-3).

* MRC (utils/mrc.py:108-144 parse, utils/loader.py:49-59 load_mrc): 1024-byte header (+ `next`
  extended bytes), modes 0/1/2/6 -> int8/int16/float32/uint16, first nz*ny*nx values reshaped
  (nz, ny, nx) and squeezed when nz == 1; then min-max normalised to [0,1] in float32
  (cv2.normalize NORM_MINMAX restated, see minmax_uint8), times 255, truncated to uint8.
* PNG/JPEG (utils/loader.py:71-93): 8-bit values "unquantised" to float32 x*6/255 - 3
  (utils/image.py:269-273); these stay float32 and are *not* divided by 255 later.
* anything else (TIFF) is taken as PIL decodes it (utils/loader.py:61-69).

The returned array is [rows, cols] = (PIL height, width).  What ``torchvision.to_tensor`` did next
(uint8 -> float32 / 255, float32 unchanged) happens on the device in the patch feed
(csrc/feed.hip) or in ``to_unit_float`` for whole-micrograph evaluation."""
import os
import struct
from collections import namedtuple

import numpy as np

_FIELDS = ("nx ny nz mode nxstart nystart nzstart mx my mz xlen ylen zlen alpha beta gamma mapc mapr maps "
           "amin amax amean ispg next creatid nint nreal imodStamp imodFlags idtype lens nd1 nd2 vd1 vd2 "
           "tilt_ox tilt_oy tilt_oz tilt_cx tilt_cy tilt_cz xorg yorg zorg cmap stamp rms nlabl labels")
# 10 ints, 6 floats, 3 ints, 3 floats, 2 ints + short, 30 pad, 2 shorts, 20 pad, 2 ints, 6 shorts,
# 6 floats, 3 floats + cmap + stamp + rms, nlabl + 10 x 80 label bytes  = 1024 bytes
_HEADER = struct.Struct("3i" "i" "3i" "3i" "3f" "3f" "3i" "3f" "2ih" "30x" "2h" "20x" "2i" "6h" "6f" "3f4s4sf" "i800s")
MRCHeader = namedtuple("MRCHeader", _FIELDS)
_MODES = {0: np.int8, 1: np.int16, 2: np.float32, 6: np.uint16}


def parse_mrc(content):
    """bytes -> (array, header, extended_header)."""
    if len(content) < 1024:
        raise ValueError("MRC file shorter than its 1024-byte header")
    header = MRCHeader._make(_HEADER.unpack(content[:1024]))
    start = 1024 + header.next
    if header.mode not in _MODES:
        raise ValueError("Unsupported MRC mode: %d" % header.mode)
    n = header.nz * header.ny * header.nx
    array = np.frombuffer(content, dtype=_MODES[header.mode], count=-1, offset=start)[:n]
    array = array.reshape(header.nz, header.ny, header.nx)
    if header.nz == 1:
        array = array[0]
    return array, header, content[1024:start]


def write_mrc(f, array, extended_header=b""):
    """Write a float32 (mode 2) MRC the way utils/mrc.py:186-222 does; array is [nz, ny, nx]."""
    array = np.asarray(array).astype(np.float32)
    if array.ndim == 2:
        array = array[None]
    zeros = [0] * 14
    header = MRCHeader(array.shape[2], array.shape[1], array.shape[0], 2, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0,
                       1, 2, 3, float(array.min()), float(array.max()), float(array.mean()), 0,
                       len(extended_header), 0, 0, 0, *zeros, 0, 0, 0, b"\x00" * 4, b"\x00" * 4,
                       float(array.std()), 0, b"\x00" * 800)
    f.write(_HEADER.pack(*header))
    f.write(extended_header)
    f.write(array.tobytes())


def minmax_uint8(image):
    """cv2.normalize(image, alpha=0, beta=1, NORM_MINMAX, CV_32F) then (x*255).astype(uint8).
    OpenCV evaluates dst = src*scale + shift with scale = 1/(max-min), shift = -min*scale formed in
    double and applied in float32; restated in that form (cv2 itself is not in this image, so the
    last-ulp rounding of this step is not pinned by a fixture)."""
    x = np.asarray(image, dtype=np.float32)
    lo, hi = float(x.min()), float(x.max())
    scale = 1.0 / (hi - lo) if hi - lo > np.finfo(np.float64).eps else 0.0
    shift = -lo * scale
    norm = x * np.float32(scale) + np.float32(shift)
    return (norm * 255).astype(np.uint8)


def unquantize(x, mi=-3, ma=3):
    return np.asarray(x).astype(np.float32) * (ma - mi) / 255 + mi


def load_image(path):
    """-> 2-D array [rows, cols]: uint8 (MRC, 8-bit TIFF) or float32 (PNG/JPEG, float TIFF)."""
    ext = os.path.splitext(path)[1]
    if ext == ".mrc":
        with open(path, "rb") as f:
            image, _, _ = parse_mrc(f.read())
        if image.ndim != 2:
            raise ValueError("%s: expected a single 2-D micrograph, got shape %s" % (path, image.shape))
        return minmax_uint8(image)
    from PIL import Image
    with Image.open(path) as im:
        im.load()
        x = np.array(im)
    if path.endswith((".png", ".jpeg", ".jpg")):
        return unquantize(x)
    if x.dtype == np.uint8:
        return x
    return x.astype(np.float32)


def to_unit_float(image):
    """torchvision.transforms.functional.to_tensor on the PIL image: uint8 -> float32 / 255."""
    if image.dtype == np.uint8:
        return image.astype(np.float32) / np.float32(255.0)
    return image.astype(np.float32)


def read_image_table(path):
    """The image list of MicrographDataset.load_data (datasets/micrograph.py:220-235): a directory is
    scanned for .mrc/.tiff/.png files; a file is a tab-separated table with image_name and path
    columns (optional source).  -> list of (source, image_name, path) in file order."""
    if os.path.isdir(path):
        import glob
        rows = []
        for p in glob.glob(path + os.sep + "*"):
            name, ext = os.path.splitext(os.path.basename(p))
            if ext in (".mrc", ".tiff", ".png"):
                rows.append((0, name, p))
        return rows
    import pandas as pd
    table = pd.read_csv(path, sep="\t")
    sources = table["source"] if "source" in table else [0] * len(table)
    return [(s, str(n), str(p)) for s, n, p in zip(sources, table["image_name"], table["path"])]
