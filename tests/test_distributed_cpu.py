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


def test_eval_sharding_three_ranks_seven_micrographs():
    """Inference shards micrograph i -> rank i % world with no collective: over 3 ranks and 7 micrographs
    the per-rank evaluation feeds are disjoint, complete and keep dataset order; a count above the dataset length wraps
    as the reference's sequential sampler does."""
    import numpy as np
    from spr_pick_amd import feed
    from spr_pick_amd.datasets import DetectionDataset
    rng = np.random.default_rng(0)
    groups = [[(rng.integers(0, 255, size=(40, 48)).astype(np.uint8), None, np.zeros((40, 48), np.float32)) for _ in range(7)]]
    names = [["mic%d" % i for i in range(7)]]
    seen = []
    for rank in range(3):
        f = feed.MicrographFeed(groups, names, count=7, device="cpu", rank=rank, world=3)
        mine = [data[DetectionDataset.METADATA][DetectionDataset.Metadata.NAME][0] for _, data in f]
        assert mine == ["mic%d" % i for i in range(rank, 7, 3)]
        assert len(f) == len(mine)
        seen += mine
    assert sorted(seen) == sorted(names[0])
    wrapped = feed.MicrographFeed(groups, names, count=9, device="cpu", rank=0, world=1)
    assert [d[DetectionDataset.METADATA][DetectionDataset.Metadata.NAME][0] for _, d in wrapped] == \
        ["mic%d" % (i % 7) for i in range(9)]


def test_init_from_env_binds_local_rank_to_its_gpu(monkeypatch):
    """One process per GPU over RCCL: with backend "nccl" the process must select cuda:LOCAL_RANK BEFORE the process
    group is created (RCCL binds communicators to the current device), with the rendezvous on 127.0.0.1 by default;
    with gloo on a box with fewer GPUs than ranks the local index wraps."""
    from spr_pick_amd import distributed
    calls = []
    monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("RANK", "5"); monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.delenv("MASTER_ADDR", raising=False); monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.delenv("SPRK_DIST_BACKEND", raising=False)
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.append(("set_device", d)))
    monkeypatch.setattr(dist, "init_process_group", lambda **kw: calls.append(("init", kw)))
    rank, world, local = distributed.init_from_env()
    assert (rank, world, local) == (5, 8, 5)
    assert calls[0] == ("set_device", 5) and calls[1][0] == "init"
    assert calls[1][1] == {"backend": "nccl", "rank": 5, "world_size": 8}
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    # gloo rehearsal on a 1-GPU box: LOCAL_RANK 5 wraps onto device 0
    calls.clear()
    monkeypatch.setenv("SPRK_DIST_BACKEND", "gloo")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    rank, world, local = distributed.init_from_env()
    assert local == 0 and calls[0][1]["backend"] == "gloo"
