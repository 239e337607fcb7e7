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
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_bucketed_allreduce():
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
