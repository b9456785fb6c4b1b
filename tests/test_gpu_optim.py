"""MultiAdam (one launch for the whole Adam update, graph_step.py / sprk_adam_multi) against torch.optim.Adam."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(96, 96, 3, 3), (96,), (1,), (7, 5), (2, 97, 1, 1), (1030,), (3, 3, 3, 3), (48, 1, 3, 3)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]


def test_multi_adam_matches_torch_adam_and_exchanges_checkpoints():
    from spr_pick_amd import graph_step
    dev = torch.device("cuda:0")
    a, b = _params(dev, 5), _params(dev, 5)
    ref = torch.optim.Adam(a, lr=1e-3, betas=(0.9, 0.99))
    opt = graph_step.make_adam(b, lr=1e-3, betas=(0.9, 0.99))
    g = torch.Generator().manual_seed(6)
    for it in range(6):
        lr = 1e-3 * (1 + it)                   # the ramp: a device scalar, no rebuild
        for grp in ref.param_groups:
            grp["lr"] = lr
        graph_step.set_lr(opt, lr)
        for i, (pa, pb) in enumerate(zip(a, b)):
            if i == 3:                         # a parameter that never gets a gradient keeps its value and has no state
                pa.grad = pb.grad = None
                continue
            gr = torch.randn(pa.shape, generator=g).to(dev) * (0.1 if it % 2 else 3.0)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        ref.step()
        opt.step()
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=2e-6, atol=2e-7), float((pa - pb).abs().max())
    assert torch.equal(a[3], _params(dev, 5)[3])
    sd = opt.state_dict()
    assert 3 not in sd["state"] and int(sd["state"][0]["step"]) == 6
    for i, st in ref.state_dict()["state"].items():
        # torch updates the first moment as a lerp, m + (1 - b1)(g - m): another rounding of the same value
        assert torch.allclose(st["exp_avg"], sd["state"][i]["exp_avg"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(st["exp_avg_sq"], sd["state"][i]["exp_avg_sq"], rtol=1e-5, atol=1e-7)
    # checkpoints go both ways: torch.optim.Adam's state into MultiAdam and back, then one more identical step
    c = _params(dev, 5)
    for pc, pa in zip(c, a):
        pc.data.copy_(pa.data)
    opt2 = graph_step.make_adam(c, lr=1e-3, betas=(0.9, 0.99))
    opt2.load_state_dict(ref.state_dict())
    ref2 = torch.optim.Adam(a, lr=1e-3, betas=(0.9, 0.99))
    ref2.load_state_dict(opt.state_dict())
    for i, (pa, pc) in enumerate(zip(a, c)):
        gr = torch.randn(pa.shape, generator=g).to(dev)
        pa.grad, pc.grad = (None, None) if i == 3 else (gr.clone(), gr.clone())
    graph_step.set_lr(opt2, 1e-3)
    for grp in ref2.param_groups:              # a loaded state dict brings its learning rate along; the trainer sets
        grp["lr"] = 1e-3                       # the ramped rate before every step anyway
    ref2.step()
    opt2.step()
    for pa, pc in zip(a, c):
        assert torch.allclose(pa, pc, rtol=2e-6, atol=2e-7)
    with pytest.raises(RuntimeError):
        graph_step.make_adam([torch.nn.Parameter(torch.zeros(3))])
