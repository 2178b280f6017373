"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel host logic: sharding, bucketed gradient
averaging (the code path RCCL takes on the GPU box, with host tensors), scalar reduction, env bootstrap."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import saunet_amd
    from saunet_amd import dp
    r, l, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    try:
        ret[rank] = fn(rank, world, dp)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return dict(ret)


def _bucket_case(rank, world, dp):
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(s)) for s in ((7, 3), (5,), (64, 64), (3, 3, 3, 3), (1000,))]
    buckets = dp.GradientBuckets(params, bucket_mb=0.01, overlap=True)   # tiny buckets -> several of them
    assert len(buckets.buckets) >= 3
    # reverse registration order: the last parameter sits in the first bucket
    assert buckets.buckets[0][0] is params[-1]
    loss = sum(((p * (rank + 1)) ** 2).sum() for p in params)
    loss.backward()          # hooks fire per parameter and launch each bucket's all-reduce when it is complete
    buckets.finish()
    want = [2 * p.detach() * sum((r + 1) ** 2 for r in range(world)) / world for p in params]
    err = max(float((p.grad - w).abs().max()) for p, w in zip(params, want))
    # second step reuses the same flat buffers
    for p in params:
        p.grad = None
    loss = sum((p * (rank + 2)).sum() for p in params)
    loss.backward(); buckets.finish()
    want2 = sum(r + 2 for r in range(world)) / world
    err2 = max(float((p.grad - want2).abs().max()) for p in params)
    vals = dp.all_reduce_scalars(torch.tensor([float(rank), 1.0]))
    return err, err2, vals.tolist()


def test_bucketed_gradient_average_world2():
    out = _run(_bucket_case, 2)
    for rank in (0, 1):
        err, err2, vals = out[rank]
        assert err < 1e-5 and err2 < 1e-6
        assert vals == [0.5, 1.0]


def _param_bcast_case(rank, world, dp):
    torch.manual_seed(rank)                      # replicas start different ...
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    dp.broadcast_parameters(m, src=0)            # ... and end identical to rank 0
    return [float(t.double().sum()) for t in list(m.parameters()) + list(m.buffers())]


def test_broadcast_parameters_world2():
    out = _run(_param_bcast_case, 2)
    assert out[0] == out[1]


def test_shards_are_disjoint_and_equal():
    import saunet_amd
    from saunet_amd import dp
    for n, world in ((103, 8), (64, 2), (10, 4)):
        shards = [dp.shard_indices(n, r, world, epoch=3) for r in range(world)]
        assert len({len(s) for s in shards}) == 1
        flat = [i for s in shards for i in s]
        assert len(flat) == len(set(flat)) == (n // world) * world
        assert shards != [dp.shard_indices(n, r, world, epoch=4) for r in range(world)]   # reshuffled every epoch
    # without drop_last every sample is seen and the tail is padded by wrap-around
    shards = [dp.shard_indices(10, r, 4, drop_last=False) for r in range(4)]
    assert set(i for s in shards for i in s) == set(range(10)) and all(len(s) == 3 for s in shards)


def test_syncbn_statistics_merge_is_exact():
    """global statistics = all-reduce of per-rank (sum, sumsq) and a count multiplied by the world size: the
    quantities conv_bn_act reduces over RCCL (lib/nn/modules/batchnorm.py:98-139 semantics, 1/sqrt(var+eps) form)."""
    torch.manual_seed(0)
    x = torch.randn(4, 6, 5, 5, dtype=torch.float64)
    shards = x.chunk(2)
    s = sum(t.sum((0, 2, 3)) for t in shards); q = sum((t * t).sum((0, 2, 3)) for t in shards)
    n = x.numel() // 6
    mean, var = s / n, q / n - (s / n) ** 2
    assert torch.allclose(mean, x.mean((0, 2, 3))) and torch.allclose(var, x.var((0, 2, 3), unbiased=False))
