"""Host-side bookkeeping of the trainer: learning-rate ramp, wall-clock accounting, duration
formatting and running means of logged tensors.

Behavioural counterpart of spr_pick/utils/utils.py (ramp :50-69, clock :104-127, formatting
:130-152, metrics :155-204).  Checkpoints written by the reference pickle instances of its ``Metric`` /
``TrackedTime`` classes, and ``checkpoint.py`` resolves those names to the classes below, so the
*instance attribute names* the pickles carry (``total``, ``n``, ``batched``, ``collapse``;
``total``, ``last_time``) are a file-format constraint.  Everything else is this package's own code.
"""
import math
import time
from collections import OrderedDict

_UNITS = ((86400, "d"), (3600, "h"), (60, "m"), (1, "s"))


def _half_cosine(t):
    """0 -> 1 smoothly as t goes 0 -> 1."""
    return 0.5 - 0.5 * math.cos(math.pi * t)


def compute_ramped_lrate(i, iteration_count, ramp_up_fraction, ramp_down_fraction, learning_rate):
    """Learning rate at sample counter ``i`` of ``iteration_count``: half-cosine rise over the first
    ``ramp_up_fraction`` of the run, squared half-cosine fall over the last ``ramp_down_fraction``
    (both may apply at once when the two windows overlap).  Pinned by tests/golden/misc.npz."""
    scale = 1.0
    up_end = iteration_count * ramp_up_fraction
    if ramp_up_fraction > 0.0 and i <= up_end:
        scale *= _half_cosine(i / up_end)
    down_start = iteration_count * (1.0 - ramp_down_fraction)
    if ramp_down_fraction > 0.0 and i >= down_start:
        remaining = 1.0 - (i - down_start) / (iteration_count * ramp_down_fraction)
        scale *= _half_cosine(remaining) ** 2
    return learning_rate * scale


class TrackedTime:
    """Accumulates the wall-clock time between successive ``update()`` calls; ``forget()`` opens a gap
    (time until the next ``update()`` is not counted), used across checkpoint save/resume."""

    def __init__(self):
        self.total = 0
        self.last_time = None

    def update(self):
        now, before = time.time(), self.last_time
        self.last_time = now
        if before is not None:
            self.total += now - before

    def forget(self):
        self.last_time = None


def seconds_to_dhms(seconds, trim=True):
    """``3661 -> "01h01m01s"``; with ``trim`` leading units that are zero are dropped."""
    fields, rest = [], seconds
    for size, unit in _UNITS:
        count, rest = rest // size, rest % size
        if unit == "s":
            count = count + rest      # fractional seconds stay with the seconds field
        if fields or not trim or count >= 1:
            fields.append("%02d%s" % (int(count), unit))
    return "".join(fields)


class Metric:
    """Mean of a logged quantity over all samples seen since the last reset.

    ``batched``: the leading axis of every added tensor is the sample axis (its length is the number
    of samples the tensor carries); ``collapse``: the remaining axes are averaged away first.  The sum
    stays a tensor on its own device — nothing is synchronised until the value is read."""

    def __init__(self, batched=True, collapse=True):
        self.batched = batched
        self.collapse = collapse
        self.total = None
        self.n = 0

    def add(self, value, scale=1.0):
        """``scale``: a constant factor of the logged quantity (the loop logs standard deviations x 255), folded
        into the accumulation instead of a launch of its own."""
        samples = value.shape[0] if self.batched else 1
        lead = 1 if self.batched else 0
        if self.collapse and self.batched:
            # mean over the non-sample axes, then sum over the samples = sum of everything / elements per sample:
            # one reduction (none for a one-element tensor) and one add per step instead of three launches
            per = max(value.numel() // max(samples, 1), 1)
            value = value.sum() if value.numel() > 1 else value.reshape(())
            scale = scale / per
        else:
            if self.collapse and value.dim() > lead:
                value = value.mean(dim=tuple(range(lead, value.dim())))
            if self.batched:
                value = value.sum(dim=0)
        if self.n == 0 or self.total is None:
            self.total = value * scale if scale != 1.0 else value
        else:
            self.total = self.total.add(value, alpha=scale)
        self.n += samples

    def __add__(self, value):     # ``metric += tensor`` in the training loop
        self.add(value)
        return self

    def reset(self):
        self.total, self.n = None, 0

    def empty(self):
        return self.n == 0

    def accumulated(self, reset=False):
        if self.empty():
            return None
        mean = self.total / self.n
        if reset:
            self.reset()
        return mean


class MetricDict(OrderedDict):
    """Ordered name -> Metric map that creates a Metric on first access."""

    def __missing__(self, key):
        created = Metric()
        self[key] = created
        return created


def separator(cols=100):
    return cols * "#"
