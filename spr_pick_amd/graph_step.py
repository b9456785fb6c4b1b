"""The training step as HIP graphs: launch-overhead-proof execution of
``zero_grad -> run_pipeline(train) -> mean(loss).backward()`` (reference loop: train.py:329-338).

A joint step is ~440 kernel launches; enqueued one by one from Python (ctypes call + workspace allocation +
autograd bookkeeping per launch) the host needs ~17 ms per step, which is fine at 32 patches per GPU (21 ms of
GPU work) and the bottleneck at 16 (BASELINE configs[3]: 128 patches over 8 GPUs) or with 16-bit convolutions.
Here the whole forward + backward is captured once per flip axis (the pipeline flips along W or H, drawn per
step) into a HIP graph — ``torch.cuda.CUDAGraph`` is hipGraph on ROCm — and replayed: one host call per step.
Nothing in the captured region synchronises with the host or has a data-dependent shape (the PU loss is
mask-based, denoiser.PuLoss; the reparameterisation noise comes from torch's graph-safe Philox stream).

Gradients are written by the backward kernels straight into slices of ONE flat fp32 buffer (``FlatGrads``):
  * both graphs write to the same addresses, so the optimiser sees the right gradients whichever was replayed;
  * the data-parallel all-reduce runs in place on that buffer — no gather / scatter copies around the collective
    (distributed.FlatGradAllReduce copies twice);
  * the 1/world of the gradient average is folded into the loss before backward (exact for world = 2^k).
The all-reduce and the Adam update stay outside the graphs (eager): Adam is ONE launch over all parameter tensors
(``MultiAdam`` / ``sprk_adam_multi``) with the step count and the learning rate in device scalars, so the ramp
needs no re-capture and no host sync.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, distributed, ops, torch_ops
from .datasets import DetectionDataset
from .feed import PinnedRing
from .params import PipelineOutput


class FlatGrads:
    """One flat fp32 buffer holding the gradient of every parameter; ``ops`` kernels write into its slices.

    ``begin_step()`` replaces ``optimizer.zero_grad(set_to_none=True)``.  During backward every operator asks
    ``dest(param)`` for the tensor to write the parameter's gradient into and gets the parameter's slice — once per
    step; a second request in the same step (a parameter used by two operators) gets a fresh tensor, which autograd
    then accumulates into the slice as usual.  Because the slice is handed to autograd as the gradient itself,
    ``param.grad`` becomes a view of the buffer without a copy."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGrads: no parameters")
        dev = self.params[0].device
        self._layout(self.params, dev)
        self.live_numel = self.flat.numel()
        self.collectives = 0          # all-reduces issued (diagnostics / tests)

    def _layout(self, order, dev):
        total = sum(p.numel() for p in order)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, off = {}, 0
        for p in order:
            n = p.numel()
            self.views[p.data_ptr()] = self.flat[off:off + n].view(p.shape)
            off += n
        self._used = set()

    def compact(self):
        """After one backward pass: parameters that received no gradient (12 tensors of the joint model, SURVEY.md
        §8a A12) move to the tail of the buffer, so that the collective covers the live prefix only.  The gradients
        of that pass move with their parameters."""
        live = [p for p in self.params if p.grad is not None]
        dead = [p for p in self.params if p.grad is None]
        old = [p.grad for p in live]           # views of the old buffer: keep it alive until they are copied
        for p in self.params:
            p.grad = None
        self._layout(live + dead, self.flat.device)
        self.live_numel = sum(p.numel() for p in live)
        # the pass that found the live set already computed this step's gradients: move them to the new layout
        # instead of running the pass again (a second pass would update the detector's BatchNorm running averages
        # and draw the reparameterisation noise once more than the reference does in its first step)
        for p, g in zip(live, old):
            v = self.views[p.data_ptr()]
            v.copy_(g)
            p.grad = v.view(v.shape)

    def adopt_strays(self):
        """A parameter used by two operators in one step gets the SUM of their gradients from autograd — a new
        tensor, not the slice one of them wrote to.  Copy such gradients into their slice and re-point ``.grad``
        (no launch at all when every parameter has one consumer, which is the joint model's case)."""
        n = 0
        for p in self.params:
            if p.grad is not None:
                v = self.views[p.data_ptr()]
                if p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                    p.grad = v.view(v.shape)
                    n += 1
        return n

    def check_adopted(self):
        """Every live parameter's ``.grad`` must BE its slice; otherwise the flat buffer would silently hold
        stale data for the collective."""
        for p in self.params:
            if p.grad is not None and p.grad.data_ptr() != self.views[p.data_ptr()].data_ptr():
                raise _lib.SprkError("FlatGrads: the gradient of a %s parameter is not in the flat buffer"
                                     % (tuple(p.shape),))

    def begin_step(self):
        self._used.clear()
        for p in self.params:
            p.grad = None

    def handed_out(self, w):
        """Has ``w``'s slice been given to an operator in this step already?"""
        return w.data_ptr() in self._used

    def dest(self, w):
        key = w.data_ptr()
        v = self.views.get(key)
        if v is None or key in self._used or v.shape != w.shape:
            return torch.empty_like(w)
        self._used.add(key)
        return v.view(v.shape)      # a fresh tensor object: autograd adopts a gradient only if nobody else holds it

    def all_reduce(self, world):
        """SUM over ranks, in place (the loss was pre-divided by ``world``)."""
        if world > 1 or (distributed.force_collective() and dist.is_initialized()):
            dist.all_reduce(self.flat[:self.live_numel], op=dist.ReduceOp.SUM)
            self.collectives += 1

    # While the context is open the backward operators leave the final sums of their two-stage reductions (weight
    # and bias gradients: ~80 five-microsecond launches per step) pending; closing it finishes them in one launch.
    defer = True

    def __enter__(self):
        ops.set_grad_destinations(self)
        return self

    def __exit__(self, *exc):
        ops.set_grad_destinations(None)
        if exc[0] is None:
            if self.flat.is_cuda and torch_ops.pending_count(self.flat.device):
                ops.flush_reductions(self.flat)
        else:
            torch_ops.drop_pending(self.flat.device)
        return False


class MultiAdam:
    """Adam as the reference configures it (train.py:128-140: lr, betas (0.9, 0.99), eps 1e-8, no weight decay) with
    the whole update in ONE launch (``sprk_adam_multi``): torch's fused Adam walks the ~130 parameter tensors in 3
    launches of 30-50 workgroups (240 us per step, 1-2 % of it); here a device-resident table maps workgroups to
    parameter tensors (1024 elements each), the moments are two flat buffers and the step count and learning rate are
    device scalars, so nothing synchronises and the ramped rate is set with ``set_lr``.

    ``state_dict()`` / ``load_state_dict()`` use torch.optim.Adam's layout (per-parameter ``step`` / ``exp_avg`` /
    ``exp_avg_sq``), so checkpoints written with either load into the other."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.99), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("MultiAdam: no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("MultiAdam runs on the GPU only")
        self.dev, self.betas, self.eps = dev, (float(betas[0]), float(betas[1])), float(eps)
        self.lr = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        self.param_groups = [{"lr": self.lr, "betas": self.betas, "eps": self.eps, "params": self.params}]
        self._offsets, off = [], 0
        for p in self.params:
            self._offsets.append(off)
            off += (p.numel() + 3) // 4 * 4          # 16-byte aligned slices
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self._steps = [torch.zeros(1, dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.float32, device=dev)]
        self._cur = 0                 # index of the buffer holding the current step count
        self._started = set()         # indices of the parameters that have taken a step (torch creates state lazily)
        self._key = None
        self._table = None

    def _slice(self, buf, i):
        p = self.params[i]
        return buf[self._offsets[i]:self._offsets[i] + p.numel()]

    def _build(self, live):
        items = (_lib.AdamItem * len(live))()
        start = np.zeros(len(live) + 1, dtype=np.int32)
        for k, i in enumerate(live):
            p = self.params[i]
            if not (p.is_contiguous() and p.grad.is_contiguous() and p.grad.dtype == torch.float32):
                raise _lib.SprkError("MultiAdam: parameters and gradients must be contiguous fp32")
            items[k] = _lib.AdamItem(p.data_ptr(), p.grad.data_ptr(), self._slice(self.exp_avg, i).data_ptr(),
                                     self._slice(self.exp_avg_sq, i).data_ptr(), p.numel())
            start[k + 1] = start[k] + (p.numel() + 1023) // 1024
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        self._table = (raw.to(self.dev), torch.from_numpy(start).to(self.dev), len(live), int(start[-1]))
        self._started.update(live)

    def step(self):
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not live:
            return
        key = tuple((i, self.params[i].data_ptr(), self.params[i].grad.data_ptr()) for i in live)
        if key != self._key:          # first step, or the gradients moved (not in a FlatGrads run): new table
            self._build(live)
            self._key = key
        items, start, n, blocks = self._table
        src, dst = self._steps[self._cur], self._steps[1 - self._cur]
        _lib.check(_lib.lib().sprk_adam_multi(items.data_ptr(), start.data_ptr(), n, blocks, self.lr.data_ptr(), src.data_ptr(),
                                              dst.data_ptr(), self.betas[0], self.betas[1], self.eps,
                                              torch.cuda.current_stream(self.dev).cuda_stream), "sprk_adam_multi")
        self._cur = 1 - self._cur

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def state_dict(self):
        step = self._steps[self._cur].clone().reshape(())
        state = {i: {"step": step.clone(), "exp_avg": self._slice(self.exp_avg, i).clone().view(self.params[i].shape),
                     "exp_avg_sq": self._slice(self.exp_avg_sq, i).clone().view(self.params[i].shape)}
                 for i in sorted(self._started)}
        return {"state": state, "param_groups": [{"lr": float(self.lr), "betas": self.betas, "eps": self.eps,
                                                  "weight_decay": 0, "amsgrad": False,
                                                  "params": list(range(len(self.params)))}]}

    def load_state_dict(self, sd):
        groups = sd.get("param_groups", [])
        if sum(len(g["params"]) for g in groups) != len(self.params):
            raise ValueError("MultiAdam: the checkpoint's optimiser has a different number of parameters")
        steps = set()
        self._started = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, st in sd["state"].items():
            i = int(i)
            if tuple(st["exp_avg"].shape) != tuple(self.params[i].shape):
                raise ValueError("MultiAdam: moment shape mismatch for parameter %d" % i)
            self._slice(self.exp_avg, i).copy_(st["exp_avg"].reshape(-1))
            self._slice(self.exp_avg_sq, i).copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(float(st["step"]))
            self._started.add(i)
        if len(steps) > 1:
            raise ValueError("MultiAdam: parameters with different step counts (%s)" % sorted(steps))
        self._steps[self._cur].fill_(steps.pop() if steps else 0.0)
        self._key = None


def make_adam(params, lr=1e-4, betas=(0.9, 0.99)):
    """Adam as the reference configures it (train.py:128-140): one launch per step, state and learning rate on the
    device (``MultiAdam``)."""
    return MultiAdam(list(params), lr=lr, betas=betas)


def set_lr(optimizer, lr):
    for group in optimizer.param_groups:
        if torch.is_tensor(group["lr"]):
            group["lr"].fill_(float(lr))
        else:
            group["lr"] = lr


_CAPTURE_MARKERS = ("capture", "captur", "hipErrorStreamCapture", "cudaErrorStreamCapture", "operation not permitted when stream is")


def _capture_specific(e):
    """Is this exception a HIP-graph CAPTURE failure (unsupported operation under capture, invalidated or unjoined
    capture) — the only kind a stepper may answer by going eager?  SprkError (a kernel wrapper refusing its arguments)
    and every other RuntimeError are genuine errors."""
    if isinstance(e, _lib.SprkError):
        return False
    msg = str(e)
    return any(m in msg for m in _CAPTURE_MARKERS)


class _NoPrep:
    """SPRK_WPREP=0: every convolution call transforms its own weights (A/B measurements)."""
    launches = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def begin_step(self, like):
        pass


_SIDE = {}


def _side_stream(dev):
    """ONE side stream per device for every GraphedTrainStep of the process: autograd keeps a parameter's
    AccumulateGrad node (and the stream it was born on) alive as long as any earlier autograd graph or captured
    graph references it, so a second stepper on the same model (bench.py's bf16 leg) must use the same stream."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(dev)
    return _SIDE[key]


class GraphedTrainStep:
    """``step(inp, target) -> outputs`` with the forward + backward replayed from a HIP graph.

    outputs: the pipeline's dictionary (static tensors, overwritten by the next call — read or clone what must
    survive).  Gradients are in ``self.grads.flat`` / ``param.grad`` afterwards; the caller runs
    ``self.grads.all_reduce(world)`` and the optimiser.  ``graph=False`` runs the same step eagerly (same kernels,
    same flat gradients) — the path the roofline leg of bench.py brackets with events."""

    def __init__(self, denoiser, batch, patch, alpha, tau, world=1, mode="joint", graph=True, eager_warmup=2,
                 draw_eps=True):
        self.den = denoiser
        self.dev = denoiser.device
        self.alpha, self.tau, self.world, self.mode = alpha, tau, world, mode
        self.use_graph = graph
        self.grads = FlatGrads(denoiser.parameters())
        self.prep = ops.WeightPrep() if os.environ.get("SPRK_WPREP", "1") != "0" else _NoPrep()
        self.inp = torch.zeros(batch, 1, patch, patch, dtype=torch.float32, device=self.dev)
        self.tgt = torch.full((batch, 1), -1.0, dtype=torch.float32, device=self.dev)
        self._tgt_ring = PinnedRing((batch, 1), torch.float32, self.dev)
        self._empty = torch.zeros(0, device=self.dev)
        # The reparameterisation noise of the two passes (joint_network_v2.py:469-475) lives in static buffers: drawn
        # inside the pass (captured: torch's graph-safe Philox stream; same draws, same order as the reference's two
        # randn_like calls), so that what a step used can be read back — or, with draw_eps=False, is whatever the
        # caller loaded (replaying a recorded step exactly, tests/test_gpu_graph_step.py).
        self.draw_eps = bool(draw_eps)
        self.eps = torch.zeros(batch, 1, patch, patch, dtype=torch.float32, device=self.dev)
        self.eps_flip = torch.zeros(batch, 1, patch, patch, dtype=torch.float32, device=self.dev)
        self.fallback_reason = None                   # set when a capture failed and the stepper went eager
        self._side = _side_stream(self.dev)           # every forward+backward runs (or is captured) on this stream
        self._graphs = {}
        self._pool = None
        self._warm = eager_warmup
        self._compacted = False
        self.kernels_per_step = None
        self._recheck = False

    # ---- one pass, eager (also what gets captured) ---------------------------------------------------
    def _pass(self, flip_p):
        with self.prep:
            # the weights changed since the last pass (the optimiser ran): every recorded weight transform of this
            # step's convolutions in one launch, the calls below then skip their own (ops.WeightPrep)
            self.prep.begin_step(self.inp)
            return self._pass_body(flip_p)

    def _pass_body(self, flip_p):
        data = DetectionDataset.make_batch(self.inp, self.tgt, hm=self._empty, hm_small=self._empty)
        if self.mode == "joint":
            if self.draw_eps:
                self.eps.normal_()
                self.eps_flip.normal_()
            o = self.den.run_pipeline(data, self.alpha, self.tau, train=True, flip_p=flip_p, eps=self.eps,
                                      eps_flip=self.eps_flip)
        else:
            o = self.den.run_pipeline(data, train=True)
        loss = torch.mean(o[PipelineOutput.LOSS])
        if self.world > 1:
            loss = loss / self.world
        loss.backward()
        self.grads.adopt_strays()
        return o

    def _eager(self, flip_p):
        # Eager passes run on the same side stream the graphs are captured on.  autograd binds a parameter's
        # AccumulateGrad node to the stream it was first used on and re-uses the node for as long as any earlier
        # autograd graph is alive (e.g. the caller still holds last step's outputs): a node born on the default
        # stream would drag that stream into a later capture (cross-stream event waits inside the captured
        # region; on ROCm the capture then dies in hipStreamEndCapture).
        cur = torch.cuda.current_stream(self.dev)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self.grads.begin_step()
            with self.grads:
                o = self._pass(flip_p)
            if not self._compacted:
                self.grads.compact()          # this pass's gradients move to the new layout with their parameters
                self._compacted = True
                self.grads.check_adopted()
            elif self._recheck:               # first eager pass after a failed capture
                self.grads.check_adopted()
                self._recheck = False
        cur.wait_stream(self._side)
        return o

    def _capture(self, axis_key, flip_p):
        L = _lib.lib()
        g = torch.cuda.CUDAGraph()
        self.grads.begin_step()
        torch.cuda.synchronize(self.dev)
        n0 = L.sprk_launch_count()
        # With a process group alive, RCCL's watchdog thread polls its events with hipEventQuery at any time; under
        # the default "global" capture mode a query from ANOTHER thread invalidates the capture.  "thread_local"
        # restricts the check to the capturing thread (the kernels the autograd thread enqueues on the capturing
        # stream are captured either way: capture is a property of the stream).
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        with torch.cuda.graph(g, pool=self._pool, stream=self._side, capture_error_mode=mode):
            with self.grads:
                o = self._pass(flip_p)
        self.kernels_per_step = L.sprk_launch_count() - n0
        self.grads.check_adopted()
        if self._pool is None:
            self._pool = g.pool()
        self._graphs[axis_key] = (g, o)

    # ---- public -----------------------------------------------------------------------------------
    def load(self, inp, target, eps=None, eps_flip=None):
        """Copy a batch into the static buffers (asynchronously; labels go through a ring of pinned buffers).
        eps / eps_flip: the reparameterisation noise of the two passes, for a stepper built with draw_eps=False."""
        if eps is not None or eps_flip is not None:
            if self.draw_eps:
                raise ValueError("this stepper draws its own noise (draw_eps=True): loaded eps would be overwritten")
            self.eps.copy_(eps, non_blocking=True)
            self.eps_flip.copy_(eps_flip, non_blocking=True)
        self.inp.copy_(inp.to(self.dev, non_blocking=True) if inp.device != self.dev else inp, non_blocking=True)
        t = target if torch.is_tensor(target) else torch.as_tensor(target)
        if t.device.type == "cuda":
            self.tgt.copy_(t.reshape(self.tgt.shape), non_blocking=True)
            return
        self._tgt_ring.upload(t.float(), out=self.tgt)

    def prepare(self, inp, target, eps=None, eps_flip=None):
        """Eager warm-up passes (lazy one-time set-up inside the library, the gradient layout) and the capture of
        both flip-axis graphs, on the given batch.  Leaves the gradients of that batch in place."""
        self.load(inp, target, eps, eps_flip)
        while self._warm > 0:
            self._warm -= 1
            self._eager(0.25)
        if not self._compacted:
            self._eager(0.25)
        if self.use_graph:
            for key, p in (("w", 0.25), ("h", 0.75)):
                if key not in self._graphs and (self.mode == "joint" or key == "w") and self.use_graph:
                    self._try_capture(key, p)

    def _try_capture(self, axis_key, flip_p):
        """Capture; if the capture raises, the stepper continues EAGERLY in the same process (same kernels, same flat
        gradients — only the host cost per step differs) and says so once.  Returns True when the graph exists."""
        try:
            self._capture(axis_key, flip_p)
            return True
        except RuntimeError as e:
            torch_ops.drop_pending(self.dev)
            ops.set_grad_destinations(None)
            if not _capture_specific(e):
                raise          # a launch failure / bad argument / device error is an error, not a performance fallback
            self.use_graph = False
            self._graphs.clear()
            self.fallback_reason = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
            import logging
            logging.getLogger(__name__).warning("HIP-graph capture of the training step failed (%s); continuing with "
                                                "eager launches", self.fallback_reason)
            try:
                torch.cuda.synchronize(self.dev)
            except RuntimeError as e2:           # the failed capture left the device in an error state: that IS fatal
                raise RuntimeError("device error after a failed HIP-graph capture: %s" % e2) from e
            # the half-captured pass may have handed out gradient slices and left partial sums: start the eager path clean
            self.grads.begin_step()
            self._recheck = True
            return False

    def __call__(self, inp, target, flip_p=None, eager=False, eps=None, eps_flip=None):
        self.load(inp, target, eps, eps_flip)
        p = float(np.random.rand()) if flip_p is None else float(flip_p)
        if eager or not self.use_graph or self._warm > 0 or not self._compacted:
            self._warm -= 1
            return self._eager(p)
        key = "w" if (p <= 0.5 or self.mode != "joint") else "h"
        if key not in self._graphs and not self._try_capture(key, 0.25 if key == "w" else 0.75):
            return self._eager(p)
        g, o = self._graphs[key]
        g.replay()
        return o
