"""Multi-process CPU test (gloo, world_size 2) of the data-parallel gradient reducer: the same bucketed
sum-all-reduce + folded 1/world mean that runs over RCCL on the GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from segmentation_training_pipeline_amd import distributed
    r, lr, w = distributed.init("gloo")
    assert (r, w) == (rank, world)
    n = 10007 * 4
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = distributed.GradReducer(bucket_mb=0.05)            # 13107-element buckets -> 4 buckets
    red.allreduce(g)
    expect = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    ok = torch.equal(g, expect) and abs(red.scale - 1.0 / world) < 1e-12 and len(red._bounds) == 4
    # bf16 wire format with host-side casts standing in for the HIP cast kernels
    g2 = torch.ones(1024) * (rank + 1)
    red2 = distributed.GradReducer(bucket_mb=0.001, wire_bf16=True,
                                   cast_fns=(lambda s, d, c: d.copy_(s.to(torch.bfloat16)), lambda s, d, c: d.copy_(s.to(torch.float32))))
    red2.allreduce(g2)
    ok = ok and torch.equal(g2, torch.ones(1024) * 3)
    # overlapped form: asynchronous per-bucket reductions issued in backward (tail-first) order, waited before the optimizer
    g3 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red3 = distributed.GradReducer(bucket_mb=0.05)
    works = [red3.allreduce_range(g3, s, e) for s, e in reversed(red3.bounds(n))]
    for wk in works:
        wk.wait()
    ok = ok and torch.equal(g3, expect)
    # ... and the overlapped form in the bf16 wire format (cast in, asynchronous reduce, cast back at wait)
    g4 = torch.arange(2048, dtype=torch.float32).remainder(64) * (rank + 1)       # small integers: exact in bf16
    red4 = distributed.GradReducer(bucket_mb=0.002, wire_bf16=True,
                                   cast_fns=(lambda s, d, c: d.copy_(s.to(torch.bfloat16)), lambda s, d, c: d.copy_(s.to(torch.float32))))
    works = [red4.allreduce_range(g4, s, e) for s, e in reversed(red4.bounds(2048))]
    for wk in works:
        wk.wait()
    ok = ok and len(works) == 4 and torch.equal(g4, torch.arange(2048, dtype=torch.float32).remainder(64) * 3)
    # error model of the bf16 wire (make_reducer's docstring) on the overlapped path, against the fp32 wire: each contribution is
    # rounded once (unit roundoff 2^-8) and the two-rank sum once more: |error| <= 2^-8 (|g_0| + |g_1|) + 2^-8 |sum| <= 2^-7 (|g_0| + |g_1|)
    gen = torch.Generator().manual_seed(100 + rank)
    g5 = torch.randn(4096, generator=gen) * 1e-3
    exact = g5.clone()
    dist.all_reduce(exact)
    mag = g5.abs()                                      # sum of the contributions' magnitudes: what the roundings scale with
    dist.all_reduce(mag)
    red5 = distributed.GradReducer(bucket_mb=0.004, wire_bf16=True,
                                   cast_fns=(lambda s, d, c: d.copy_(s.to(torch.bfloat16)), lambda s, d, c: d.copy_(s.to(torch.float32))))
    for wk in [red5.allreduce_range(g5, s, e) for s, e in reversed(red5.bounds(4096))]:
        wk.wait()
    ok = ok and bool(((g5 - exact).abs() <= 2.0 ** -7 * mag + 1e-12).all()) and not torch.equal(g5, exact)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_bucketed_allreduce():
    """(also the bf16-wire error model: test_bf16_wire_error_model is an alias of this multi-process run)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_overlap_schedule_covers_every_bucket_once_in_backward_order():
    """distributed.overlap_schedule on the real plan of U-Net/ResNet34: segments end at increasing launch indices,
    each bucket appears once, and a bucket is only released when no later backward launch can write into it."""
    from segmentation_training_pipeline_amd import distributed, graph, nets
    plan = graph.Plan(batch=2, dtype="bf16", device="cpu")
    plan.define(lambda p: nets.unet_resnet(p, "resnet34", 64, 64))
    assert plan.bwd_monotone
    ends = [e for e, _ in plan.bwd_marks]
    lows = [l for _, l in plan.bwd_marks]
    assert ends == sorted(ends) and ends[-1] == len(plan.bwd) and lows == sorted(lows, reverse=True) and lows[-1] == 0
    n = plan.G.numel()
    bounds = distributed.bucket_bounds(n, 4 * (1 << 20))      # 16 MB buckets -> 6 buckets
    segs = distributed.overlap_schedule(plan.bwd_marks, bounds, len(plan.bwd))
    assert [e for e, _ in segs] == sorted(set(e for e, _ in segs)) and segs[-1][0] == len(plan.bwd)
    seen = [r for _, rs in segs for r in rs]
    assert sorted(seen) == bounds and len(segs) >= 4
    for end, rs in segs[:-1]:
        low_at_end = [l for e, l in plan.bwd_marks if e == end][-1]
        assert all(s >= low_at_end for s, _ in rs)
    assert (0, bounds[0][1]) in segs[-1][1]
    # default schedule: two phases, the tail (> 90 % of the bytes) final at 70 % of the launches
    two = distributed.two_phase_bounds(plan.bwd_marks, n)
    assert len(two) == 2 and two[0][0] == 0 and two[1][1] == n and two[0][1] == two[1][0] and two[0][1] < 0.1 * n
    s2 = distributed.overlap_schedule(plan.bwd_marks, two, len(plan.bwd))
    assert len(s2) == 2 and 0.6 * len(plan.bwd) < s2[0][0] < 0.85 * len(plan.bwd) and s2[1][0] == len(plan.bwd)
    # a frozen encoder never reaches the head of the arena: everything left goes with the last segment
    short = [(e, max(l, n // 2)) for e, l in plan.bwd_marks]
    segs2 = distributed.overlap_schedule(short, bounds, len(plan.bwd))
    assert sorted(r for _, rs in segs2 for r in rs) == bounds and segs2[-1][0] == len(plan.bwd)


def _epoch_worker(rank, world, port, q):
    """Rank-consistent epoch bookkeeping (pipeline.reduce_epoch_sums / distributed.allreduce_sums / broadcast_tensors): the
    ranks feed DIFFERENT shard sums and must come out with the SAME logs bit for bit - which is what makes the best-checkpoint
    choice, EarlyStopping and ReduceLROnPlateau identical on every rank (ranks could stop alone and hang)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from segmentation_training_pipeline_amd import distributed, pipeline
    distributed.init("gloo")
    names = pipeline.epoch_log_names(1)
    # rank 0 saw 5 samples, rank 1 saw 3 (validation shards differ by design); rank-specific metric sums
    n = 5 if rank == 0 else 3
    sums = {k: (i + 1) * 0.37 * n * (1.0 + 0.1 * rank) for i, k in enumerate(names)}
    logs = pipeline.reduce_epoch_sums(sums, n, 1)
    expect = {k: ((i + 1) * 0.37 * 5 * 1.0 + (i + 1) * 0.37 * 3 * 1.1) / 8.0 for i, k in enumerate(names)}
    ok = all(abs(logs[k] - expect[k]) < 1e-12 for k in names)
    # an empty shard still takes part in the collective (fixed vector layout)
    logs2 = pipeline.reduce_epoch_sums({} if rank == 1 else {k: 2.0 * 4 for k in names}, 0 if rank == 1 else 4, 1)
    ok = ok and all(abs(logs2[k] - 2.0) < 1e-12 for k in names)
    # callbacks driven by the reduced logs take the same decision everywhere
    es = pipeline.EarlyStopping(patience=1, monitor="val_loss")
    for ep, v in enumerate((1.0, 0.9, 0.95)):
        es.on_epoch_end(None, ep, {"val_loss": pipeline.reduce_epoch_sums({"loss": v * n * (1 + rank)}, n, 1)["loss"]})
    # parameter broadcast: rank 1 starts from garbage and ends with rank 0's tensors
    t = torch.arange(1000, dtype=torch.float32) if rank == 0 else torch.full((1000,), -7.0)
    s = torch.ones(10) * (3.0 if rank == 0 else 9.0)
    distributed.broadcast_tensors([t, None, s], src=0)
    ok = ok and torch.equal(t, torch.arange(1000, dtype=torch.float32)) and torch.equal(s, torch.ones(10) * 3.0)
    avg = torch.ones(4) * (rank + 1)
    distributed.average_tensor(avg)
    ok = ok and torch.equal(avg, torch.ones(4) * 1.5)
    sh = distributed.shard_list(list(range(7)), rank, world)
    q.put((rank, bool(ok), tuple(sorted(logs.items())), es.stop, tuple(sh)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_epoch_scalars_and_decisions_are_rank_consistent():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_epoch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[:2] for r in res] == [(0, True), (1, True)]
    assert res[0][2] == res[1][2]                       # identical logs, bit for bit
    assert res[0][3] == res[1][3] is True                # the same stop decision
    assert sorted(res[0][4] + res[1][4]) == list(range(7))   # validation shards partition the set


test_bf16_wire_error_model = test_gloo_world2_bucketed_allreduce
