"""Image / score files of the validation and evaluation loops, written off the critical path.

The reference writes six PNGs and one score table per micrograph synchronously inside its evaluation loop
(train.py:500-585, utils/data.py:71-93,143-147 ``save_tensor_image``: min-max normalise the [1,a,b] tensor,
swap the two image axes back to the file's orientation, quantise with ``uint8(x*255)``).  At the rate the
filled network runs here (a 1024^2 micrograph in ~20 ms) that host work — 4 MB device-to-host copies and
~40 ms of deflate per image — would be 10x the GPU time.  Here:

* normalisation, transpose and quantisation run on the device (same IEEE float32 operations in the same order
  as the NumPy statement, so the bytes are identical — tests/test_gpu_trainer.py), 1 byte per pixel crosses PCIe
  into a pinned buffer, asynchronously;
* a small thread pool waits for the copy's event, deflates (zlib releases the GIL) and writes the file.
  The encoder emits plain 8-bit greyscale PNGs (filter 0, one IDAT) — other bytes than PIL's encoder, the same pixels.

``drain()`` blocks until every file is on disk; the loops call it before they report."""
import os
import struct
import threading
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def png_bytes(img, level=6):
    """uint8 [rows, cols] -> bytes of an 8-bit greyscale PNG."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    raw = np.empty((h, w + 1), dtype=np.uint8)
    raw[:, 0] = 0                       # filter type 0 (None) in front of every scanline
    raw[:, 1:] = img

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + chunk(b"IEND", b""))


def quantise(img):
    """save_tensor_image's arithmetic on whatever device `img` [1,a,b] lives on -> uint8 [b,a] (transposed back to the
    file's orientation).  float32 throughout: (x - lo) / (hi - lo) * 255, truncated; a constant image gives zeros."""
    x = img.detach().to(torch.float32)[0]
    lo, hi = x.min(), x.max()
    span = hi - lo
    y = torch.where(span > 0, (x - lo) / torch.where(span > 0, span, torch.ones_like(span)), torch.zeros_like(x))
    return (y.t() * 255).to(torch.uint8).contiguous()


class OutputWriter:
    def __init__(self, threads=None):
        n = threads or int(os.environ.get("SPRK_WRITER_THREADS", "0")) or max(2, min(12, (os.cpu_count() or 4) - 2))
        self.pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="sprk-out")
        self._pending = []
        self._lock = threading.Lock()
        self._free = []                # pinned staging buffers by size

    def _staging(self, numel):
        with self._lock:
            for k, b in enumerate(self._free):
                if b.numel() == numel:
                    return self._free.pop(k)
        return torch.empty(numel, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else torch.empty(numel, dtype=torch.uint8)

    def _submit(self, fn, *a):
        fut = self.pool.submit(fn, *a)
        self._pending.append(fut)
        if len(self._pending) > 256:       # bound the pinned memory and the queue: wait for the oldest
            self._pending.pop(0).result()

    def png(self, img, path):
        """img: tensor [1,a,b] (any float dtype, any device)."""
        q = quantise(img)
        if q.is_cuda:
            buf = self._staging(q.numel())
            buf.view(q.shape).copy_(q, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(q.device))
            self._submit(self._encode_staged, buf, tuple(q.shape), ev, path)
        else:
            self._submit(self._encode, q.numpy(), path)

    def _encode_staged(self, buf, shape, ev, path):
        ev.synchronize()
        self._encode(buf.view(shape).numpy(), path)
        with self._lock:
            if len(self._free) < 64:
                self._free.append(buf)

    @staticmethod
    def _encode(arr, path):
        data = png_bytes(arr)
        with open(path, "wb") as f:
            f.write(data)

    def call(self, fn, *a):
        """Any other file job (score tables), in submission order relative to nothing: jobs are independent."""
        self._submit(fn, *a)

    def drain(self):
        pending, self._pending = self._pending, []
        for fut in pending:
            fut.result()               # re-raises a worker's exception here


def tensor_to_png(img, path):
    """Synchronous form (one image): quantise + encode + write."""
    OutputWriter._encode(quantise(img).cpu().numpy(), path)
