"""world_size-2 gloo test of the N>1 path: trajectory sharding + metric reduction (no kernels)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from magnet_amd.sharding import allreduce_gradients, gather_predictions, max_over_ranks, shard_batch, shard_range, sum_over_ranks


def test_shard_range_partitions_everything():
    for n in (0, 1, 5, 8, 13):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in shard_range(n, r, w)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = {"u": torch.arange(10.0)[:, None].repeat(1, 3), "t": torch.arange(10.0)[:, None]}
    mine = shard_batch(batch, rank, world)
    pred = mine["u"] * 2                                   # stand-in for a per-trajectory rollout
    full = gather_predictions(pred)
    ok = torch.equal(full, batch["u"] * 2)
    tmax = max_over_ranks(1.0 + rank)
    tot = sum_over_ranks(float(mine["u"].shape[0]))
    out[rank] = (ok, tmax, tot)
    dist.destroy_process_group()


def test_two_rank_shard_and_reduce():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    for r in (0, 1):
        ok, tmax, tot = out[r]
        assert ok and tmax == 2.0 and tot == 10.0


def _grad_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))   # same init on all ranks
    x, y = torch.randn(4, 6), torch.randn(4, 3)
    full = torch.nn.functional.l1_loss(net(x), y)
    want = torch.autograd.grad(full, list(net.parameters()))
    mine = shard_batch({"x": x, "y": y}, rank, world)
    torch.nn.functional.l1_loss(net(mine["x"]), mine["y"]).backward()
    allreduce_gradients(net.parameters())
    out[rank] = [p.grad.clone() for p in net.parameters()], [w.clone() for w in want]
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_full_batch_gradient():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_grad_worker, args=(2, port, out), nprocs=2, join=True)
    g0, want = out[0]
    g1, _ = out[1]
    for a, b, w in zip(g0, g1, want):
        assert torch.equal(a, b)                                   # identical on every rank
        assert torch.allclose(a, w, rtol=1e-5, atol=1e-7)          # mean of shard gradients == full-batch gradient
