"""CPU tests of the host logic around the graph-captured training step (spr_pick_amd/graph_step.py): the flat
gradient buffer the backward kernels write into (adoption by autograd without a copy, the accumulate fallback,
compaction of never-used parameters), its in-place all-reduce over gloo at world sizes 2 and 4 with the joint
model's real parameter shapes, and the mask-based (host-sync-free) PU loss against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

NEVER_GRAD = ("denoiser_model.denoise_branch.output_conv_f.", "sigma_estimation_model.decode_block_3.",
              "sigma_estimation_model.detect_block.", "sigma_estimation_model.output_conv_f.")


class _Linear(torch.autograd.Function):
    """y = x @ w.T with the weight gradient written where ops._grad_like says (as the HIP operators do)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, gy):
        from spr_pick_amd import ops
        x, w = ctx.saved_tensors
        gw = ops._grad_like(w)
        torch.mm(gy.t(), x, out=gw)
        return gy @ w, gw


def test_flat_grads_adoption_and_accumulate_fallback():
    from spr_pick_amd.graph_step import FlatGrads
    torch.manual_seed(0)
    w1, w2, w3 = (torch.nn.Parameter(torch.randn(s)) for s in ((4, 3), (2, 4), (5, 5)))
    fg = FlatGrads([w1, w2, w3])
    x = torch.randn(6, 3)

    def run(twice):
        fg.begin_step()
        with fg:
            h = _Linear.apply(x, w1)
            if twice:
                h = h + _Linear.apply(2 * x, w1)      # the same parameter used by two operators in one step
            _Linear.apply(h, w2).sum().backward()
        return fg.adopt_strays()

    assert run(False) == 0
    fg.check_adopted()
    assert w1.grad.data_ptr() == fg.views[w1.data_ptr()].data_ptr()      # .grad IS the slice: no copy
    assert w3.grad is None
    ref = torch.autograd.grad((x @ w1.t() @ w2.t()).sum(), [w1, w2])
    assert torch.allclose(w1.grad, ref[0]) and torch.allclose(w2.grad, ref[1])
    fg.compact()                                                           # w3 never got a gradient: tail
    assert fg.live_numel == w1.numel() + w2.numel() and fg.flat.numel() == fg.live_numel + 25
    assert run(True) == 1          # autograd summed two gradients of w1 into a new tensor: copied into the slice
    fg.check_adopted()
    ref = torch.autograd.grad(((x @ w1.t() + 2 * x @ w1.t()) @ w2.t()).sum(), [w1, w2])
    assert torch.allclose(w1.grad, ref[0], atol=1e-5) and torch.allclose(w2.grad, ref[1], atol=1e-5)
    flat_w1 = fg.flat[:12].view(4, 3) if fg.views[w1.data_ptr()].data_ptr() == fg.flat.data_ptr() else None
    assert flat_w1 is None or torch.equal(flat_w1, w1.grad)
    # outside the context manager the operators allocate as before
    from spr_pick_amd import ops
    assert ops._GRAD_DEST is None and ops._grad_like(w1).data_ptr() != fg.views[w1.data_ptr()].data_ptr()


def test_pu_loss_mask_form_matches_oracle_on_every_label_mix():
    """denoiser.PuLoss (device-side counts, fixed shapes) against the oracle's statement of utils/losses.py:303-349
    for mixed, all-unlabelled, all-labelled and single-unlabelled batches."""
    from oracle import pipeline
    from spr_pick_amd.denoiser import PuLoss
    g = torch.Generator().manual_seed(3)
    pu = PuLoss()
    for B, labels in ((16, None), (16, "none"), (8, "all"), (8, "one_unl"), (32, None)):
        p = torch.rand(B, 1, 1, 1, generator=g) * 0.98 + 0.01
        y = torch.where(torch.rand(B, 1, generator=g) < 0.3, torch.rand(B, 1, generator=g), torch.full((B, 1), -1.0))
        if labels == "none":
            y = torch.full((B, 1), -1.0)
        elif labels == "all":
            y = torch.rand(B, 1, generator=g)
        elif labels == "one_unl":
            y = torch.rand(B, 1, generator=g)
            y[3] = -1.0
        for tau in (0.01, 0.2):
            want = pipeline.pu_loss(tau, p, y)
            got = pu.mask_form(tau, p, y)
            assert torch.allclose(got, want, rtol=2e-5, atol=1e-6), (B, labels, tau, float(got), float(want))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from oracle import weights
    from spr_pick_amd import distributed
    from spr_pick_amd.graph_step import FlatGrads
    r, w, _ = distributed.init_from_env(backend="gloo")
    shapes = weights.denoiser_shapes()
    names = [k for k in shapes if "running_" not in k and "num_batches" not in k]
    params = [torch.nn.Parameter(torch.zeros(shapes[k])) for k in names]
    dead = [any(k.startswith(pre) for pre in NEVER_GRAD) for k in names]
    assert sum(dead) == 12 and sum(p.numel() for p in params) == 2333320
    fg = FlatGrads(params)

    def fake_backward(step):
        fg.begin_step()
        for i, (p, d) in enumerate(zip(params, dead)):
            if not d:
                v = fg.dest(p)
                v.fill_(float((rank + 1) * (i % 7 + 1) + step))
                p.grad = v
    fake_backward(0)
    fg.compact()
    assert fg.live_numel == 2106950          # the all-reduce payload
    for step in (1, 2):
        fake_backward(step)
        fg.check_adopted()
        fg.all_reduce(w)
        for i, (p, d) in enumerate(zip(params, dead)):
            if d:
                assert p.grad is None
            else:
                want = sum((rr + 1) * (i % 7 + 1) + step for rr in range(w))
                assert float(p.grad.min()) == want == float(p.grad.max()), (i, step)
        assert float(fg.flat[fg.live_numel:].abs().max()) == 0.0      # never-grad tail is outside the collective
    out.put(rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_flat_grads_allreduce_real_shapes(world):
    ctx = mp.get_context("spawn")
    port, q = _free_port(), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(world)) == list(range(world))
