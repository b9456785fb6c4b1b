"""Small host utilities of the trainer, restating spr_pick/utils/utils.py: the learning-rate ramp
(:50-69), time tracking (:104-127), duration formatting (:130-152), metric accumulation
(:155-204) and the log separator (:207-208)."""
import time
from collections import OrderedDict

import numpy as np
import torch


def compute_ramped_lrate(i, iteration_count, ramp_up_fraction, ramp_down_fraction, learning_rate):
    """Cosine ramp-up over the first `ramp_up_fraction` of the run and squared-cosine ramp-down over
    the last `ramp_down_fraction`."""
    if ramp_up_fraction > 0.0:
        if i <= iteration_count * ramp_up_fraction:
            t = (i / ramp_up_fraction) / iteration_count
            learning_rate = learning_rate * (0.5 - np.cos(t * np.pi) / 2)
    if ramp_down_fraction > 0.0:
        start = iteration_count * (1 - ramp_down_fraction)
        if i >= start:
            t = ((i - start) / ramp_down_fraction) / iteration_count
            learning_rate = learning_rate * (0.5 + np.cos(t * np.pi) / 2) ** 2
    return learning_rate


class TrackedTime:
    def __init__(self):
        self.total = 0
        self.last_time = None

    def update(self):
        now = time.time()
        if self.last_time is not None:
            self.total += now - self.last_time
        self.last_time = now

    def forget(self):
        self.last_time = None


def seconds_to_dhms(seconds, trim=True):
    parts = [(seconds // 86400, "d"), (seconds // 3600 % 24, "h"), ((seconds // 60) % 60, "m"), (seconds % 60, "s")]
    out = ""
    for value, unit in parts:
        if trim and value < 1:
            continue
        trim = False
        out += "{:02}{}".format(int(value), unit)
    return out


class Metric:
    """Running mean over the batch axis; values stay on their device until read."""

    def __init__(self, batched=True, collapse=True):
        self.batched, self.collapse = batched, collapse
        self.reset()

    def add(self, value):
        n = value.shape[0] if self.batched else 1
        if self.collapse:
            dims = list(range(1 if self.batched else 0, value.dim()))
            if dims:
                value = torch.mean(value, dim=dims)
        if self.batched:
            value = torch.sum(value, dim=0)
        self.total = value if self.total is None else self.total + value
        self.n += n

    def __add__(self, value):
        self.add(value)
        return self

    def accumulated(self, reset=False):
        if self.n == 0:
            return None
        acc = self.total / self.n
        if reset:
            self.reset()
        return acc

    def reset(self):
        self.total = None
        self.n = 0

    def empty(self):
        return self.n == 0


class MetricDict(OrderedDict):
    def __missing__(self, key):
        self[key] = value = Metric()
        return value


def separator(cols=100):
    return "#" * cols
