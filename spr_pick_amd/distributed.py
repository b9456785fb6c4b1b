"""Data-parallel training support: one process per GPU, gradients averaged with ONE flat
all-reduce per step over RCCL/xGMI (backend "nccl" on ROCm; "gloo" for the CPU tests).

The reference has no distributed code; the joint step shards naturally over
patches: every rank draws its own patches, runs the full step locally, and the 2.1 M fp32
gradients (8.4 MB) are summed in a single collective — latency-bound on xGMI, so no bucketing.
Parameters that never receive a gradient (12 tensors) stay ``grad=None`` on
every rank and are left out of the buffer, exactly as Adam skips them in the reference.
BatchNorm statistics and the PU-loss counts stay per-rank.
"""
import os

import torch
import torch.distributed as dist


def force_collective():
    """SPRK_DIST_FORCE=1: create the process group and run the gradient collective even at world size 1 — the way to
    make RCCL initialisation, the collective on the flat gradient buffer and HIP-graph capture beside RCCL's watchdog
    thread meet each other on a one-GPU box (tests/test_gpu_trainer.py).  A 1-rank all-reduce is the identity."""
    return os.environ.get("SPRK_DIST_FORCE", "0") == "1"


def backend_name():
    """Name of the active collective library for reports: "RCCL" (torch backend "nccl" on ROCm), "gloo", or None."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    b = dist.get_backend()
    return "RCCL" if b == "nccl" else b


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    Returns (rank, world_size, local_rank); a no-op for single-process runs (unless SPRK_DIST_FORCE=1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_collective()) and not dist.is_initialized():
        if backend is None:
            # SPRK_DIST_BACKEND=gloo: rehearse the multi-process path on a box with fewer GPUs than ranks
            backend = os.environ.get("SPRK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            local = local % torch.cuda.device_count()
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGradAllReduce:
    """Average ``.grad`` of the given parameters across ranks with one collective."""

    def __init__(self, params, world_size=None):
        self.params = list(params)
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self._flat = None
        self._live = None
        self._views = None

    def _select(self):
        live = [p for p in self.params if p.grad is not None]
        sig = tuple(id(p) for p in live)
        if self._live is None or self._sig != sig:
            n = sum(p.grad.numel() for p in live)
            self._flat = torch.empty(n, dtype=live[0].grad.dtype, device=live[0].grad.device) if live else None
            self._live, self._sig = live, sig
            self._views = None
        return self._live

    def numel(self):
        return 0 if self._flat is None else self._flat.numel()

    def __call__(self):
        if self.world <= 1:
            return
        live = self._select()
        if not live:
            return
        if self._views is None:
            off, self._views = 0, []
            for p in live:
                n = p.grad.numel()
                self._views.append(self._flat[off:off + n])
                off += n
        # two multi-tensor copies instead of one small kernel per parameter on each side of the collective
        grads = [p.grad.reshape(-1) for p in live]
        torch._foreach_copy_(self._views, grads)
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
        self._flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, self._views)
        for p, g in zip(live, grads):
            if g.data_ptr() != p.grad.data_ptr():      # non-contiguous gradient: reshape(-1) made a copy
                p.grad.copy_(g.view_as(p.grad))


def shard_indices(n_items, rank, world):
    """Units (micrographs) owned by ``rank``: item i goes to rank i % world."""
    return list(range(rank, n_items, world))
