"""CPU, world_size 2 over gloo: the flat-gradient all-reduce used for data-parallel training and
the micrograph sharding rule used for inference."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from spr_pick_amd import distributed
    r, w, _ = distributed.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    # same weights on every rank; three parameters, one of which never receives a gradient
    model = torch.nn.ModuleDict({"a": torch.nn.Linear(5, 3), "b": torch.nn.Linear(3, 1), "unused": torch.nn.Linear(2, 2)})
    params = list(model.parameters())
    sync = distributed.FlatGradAllReduce(params, w)
    g = torch.Generator().manual_seed(100 + rank)       # per-rank data shard
    x = torch.randn(4, 5, generator=g)
    loss = model["b"](torch.relu(model["a"](x))).mean()
    loss.backward()
    local = [None if p.grad is None else p.grad.clone() for p in params]
    sync()
    gathered = [None] * w
    dist.all_gather_object(gathered, local)
    for i, p in enumerate(params):
        if p.grad is None:
            assert all(gr[i] is None for gr in gathered)
            continue
        want = sum(gr[i] for gr in gathered) / w
        assert torch.allclose(p.grad, want, rtol=0, atol=1e-7), i
    assert sync.numel() == sum(p.numel() for p in list(model["a"].parameters()) + list(model["b"].parameters()))
    # a second step re-uses the flat buffer
    for p in params:
        p.grad = None
    model["b"](torch.relu(model["a"](x))).sum().backward()
    sync()
    out.put((rank, distributed.shard_indices(7, rank, w)))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2():
    mp.set_start_method("spawn", force=True)
    port = _free_port()
    q = mp.get_context("spawn").Queue()
    procs = [mp.get_context("spawn").Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    shards = dict(q.get() for _ in range(2))
    assert shards[0] == [0, 2, 4, 6] and shards[1] == [1, 3, 5]


def test_single_process_is_a_noop():
    from spr_pick_amd import distributed
    lin = torch.nn.Linear(2, 2)
    lin(torch.ones(1, 2)).sum().backward()
    before = lin.weight.grad.clone()
    distributed.FlatGradAllReduce(lin.parameters(), 1)()
    assert torch.equal(before, lin.weight.grad)
