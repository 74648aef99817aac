"""Data-parallel gradient reduction on CPU with gloo, world_size 2: bucket planning covers the arena
exactly once, buckets are released in backward order, and the reduced arena equals the rank sum."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dspnet_amd.train.solver import GradBucketReducer, plan_buckets


def fake_layout(n_params=23, seed=0):
    rng = np.random.default_rng(seed)
    params, owner, off = [], {}, 0
    for i in range(n_params):
        size = int(rng.integers(1, 4000)) // 4 * 4 + 4
        name = "p%d" % i
        params.append((name, off, size))
        owner[name] = i // 2          # two params per node, in forward order
        off += size
    return params, owner, off


def test_plan_buckets_partition_and_order():
    params, owner, total = fake_layout()
    buckets = plan_buckets(params, owner, total, 6000)
    spans = sorted((lo, hi) for lo, hi, _ in buckets)
    assert spans[0][0] == 0 and spans[-1][1] == total
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c                              # contiguous, no overlap, no gap
    firsts = [f for _, _, f in buckets]
    assert firsts == sorted(firsts, reverse=True)  # released from the end of the network first
    for lo, hi, first in buckets:                  # a bucket is released only after all its owners ran
        owners = [owner[n] for n, o, s in params if o < hi and o + s > lo]
        assert first == min(owners)
    assert len(buckets) > 3


def real_layout():
    """the gradient-arena layout of the headline graph (resnet-50 multitask 512x512), exported on the MI355X box by
    tests/golden/make_bucket_layout_golden.py: 211 parameters, 32.4 M floats, 292 graph nodes"""
    import json
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bucket_layout_resnet50_512.json")))
    params = [(n, o, s) for n, o, s, _ in doc["params"]]
    owner = {n: i for n, _, _, i in doc["params"]}
    return params, owner, doc["arena"], doc


def test_plan_buckets_on_the_real_graph_layout():
    """the bucket plan of the REAL graph: covers the arena exactly once, 16 MB buckets -> the 8 collectives bench.py reports,
    released in backward order (decoder / heads first, conv0 last), each only after every gradient in it is final; and it
    is the plan the device run itself made (recorded next to the layout)"""
    params, owner, total, doc = real_layout()
    assert len(params) > 200 and total > 32e6
    buckets = plan_buckets(params, owner, total, int(16.0 * (1 << 20) / 4))
    assert [list(b) for b in buckets] == doc["buckets_16mb"]
    spans = sorted((lo, hi) for lo, hi, _ in buckets)
    assert spans[0][0] == 0 and spans[-1][1] == total and all(b == c for (_, b), (c, _) in zip(spans, spans[1:]))
    assert len(buckets) == 8
    firsts = [f for _, _, f in buckets]
    assert firsts == sorted(firsts, reverse=True) and firsts[-1] <= 3          # the last bucket waits for conv0 / bn_data
    for lo, hi, first in buckets:
        assert first == min(owner[n] for n, o, s in params if o < hi and o + s > lo)
    # the arena is in forward order: the bucket released first holds the parameters of the LAST layers
    assert max(buckets, key=lambda b: b[2])[1] == total


def _worker(rank, world, port, total, params, owner, out, bucket_elems=6000):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    arena = torch.randn(total, generator=g)
    expect_local = arena.clone()
    buckets = plan_buckets(params, owner, total, bucket_elems)
    red = GradBucketReducer(arena, buckets)
    red.begin()
    n_nodes = max(owner.values()) + 1
    released_at = {}
    for idx in range(n_nodes - 1, -1, -1):       # "backward": last node first
        before = len(red.launched)
        red.node_done(idx)
        for span in red.launched[before:]:
            released_at[span] = idx
    red.finish()
    gathered = [torch.zeros(total) for _ in range(world)]
    dist.all_gather(gathered, expect_local)
    ok = torch.allclose(arena, sum(gathered), rtol=0, atol=1e-5)
    order_ok = all(released_at[(lo, hi)] == first for lo, hi, first in buckets)
    out.put((rank, bool(ok), bool(order_ok), len(red.launched)))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    params, owner, total = fake_layout()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, params, owner, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, order_ok, n in res:
        assert ok and order_ok and n > 3, (rank, ok, order_ok, n)


def test_bucketed_allreduce_world2_gloo_on_the_real_graph_layout():
    """world 2, gloo, the REAL arena (129 MB of float32 per rank) and the bucket size of the GPU run: every bucket is
    released at the node the plan names, exactly once, and the reduced arena is the rank sum"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    params, owner, total, _ = real_layout()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, params, owner, q, int(16.0 * (1 << 20) / 4))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, order_ok, n in res:
        assert ok and order_ok and n == 8, (rank, ok, order_ok, n)


def test_side_stream_buckets_on_the_real_graph_layout():
    """Round 5 (N > 1 runs the N = 1 schedule): which buckets of the REAL graph must wait for the side stream before their
    release -- exactly those holding a parameter of a node whose backward runs there (the SSD extra layers behind the first
    and the heads on the extra maps) -- and that the reducer calls the solver's hook once per bucket, in release order,
    BEFORE the bucket's collective is issued."""
    from dspnet_amd.train.solver import side_buckets
    params, owner, total, doc = real_layout()
    buckets = plan_buckets(params, owner, total, int(4.0 * (1 << 20) / 4))
    side_nodes = {owner[n] for n, _, _ in params
                  if n.startswith("multi_feat_") and not n.startswith("multi_feat_2_conv_1x1")}
    flags = side_buckets(buckets, params, owner, side_nodes)
    assert any(flags) and not all(flags)
    for (lo, hi, _), f in zip(buckets, flags):
        names = [n for n, o, s in params if o < hi and o + s > lo]
        assert f == any(owner[n] in side_nodes for n in names)
    assert side_buckets(buckets, params, owner, ()) == [False] * len(buckets)

    class FakeWork:
        def wait(self):
            pass
    log = []
    red = GradBucketReducer(torch.zeros(total), buckets)
    import torch.distributed as dist_mod
    real = dist_mod.all_reduce
    dist_mod.all_reduce = lambda t, group=None, async_op=False: (log.append(("reduce", t.numel())), FakeWork())[1]
    try:
        red.begin()
        for idx in range(doc["nodes"] - 1, -1, -1):
            red.node_done(idx, lambda b: log.append(("hook", b)))
        red.finish()
    finally:
        dist_mod.all_reduce = real
    assert [e for e in log if e[0] == "hook"] == [("hook", b) for b in range(len(buckets))]
    for k in range(len(buckets)):          # hook b directly precedes collective b
        assert log[2 * k] == ("hook", k) and log[2 * k + 1] == ("reduce", buckets[k][1] - buckets[k][0])
