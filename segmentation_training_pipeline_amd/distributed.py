"""Data-parallel training: one process per GPU, gradients averaged with RCCL over xGMI.

The reference's multi-GPU mode is ``cfg.gpus = N`` (README.md:756-760: in-graph replicas via
keras ``multi_gpu_model``, no collective library); ``--num_gpus/--gpus_per_net`` of ``musket fit``
(README.md:45-57) choose how many devices an experiment may use.  Here every rank owns a full
replica; after the backward graph the flat fp32 gradient arena is all-reduced in a few large
buckets (xGMI is point-to-point: ring collectives are per-link bound, so buckets are sized in
tens of MB, not per layer) and the 1/world mean is folded into the optimizer's gradient scale.
BatchNormalization statistics stay per replica, as with the reference's towers.

``torch.distributed`` backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU test-suite to
exercise the same bucketing logic.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def device_index(local_rank):
    """GPU of a local rank: its own device, wrapped around when the node exposes fewer devices than ranks (tests)."""
    n = torch.cuda.device_count()
    return local_rank % n if n else 0


def init(backend=None, force=False):
    """Joins the torchrun rendezvous (env://).  With world size 1 nothing is initialised unless ``force``
    (used to exercise the RCCL path on a single GPU)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend is None:
            # STP_DIST_BACKEND=gloo: several ranks sharing ONE GPU (RCCL refuses duplicate devices) - the 2-rank tests
            backend = os.environ.get("STP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(device_index(local_rank))
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def bucket_bounds(numel, bucket_elems, align=4):
    """Splits [0, numel) into contiguous buckets of ~bucket_elems (multiples of ``align``)."""
    bucket_elems = max(align, (int(bucket_elems) // align) * align)
    out, s = [], 0
    while s < numel:
        e = min(numel, s + bucket_elems)
        out.append((s, e))
        s = e
    return out


def overlap_schedule(marks, bounds, n_launches):
    """Cuts a backward launch list into segments after which gradient buckets are final.

    ``marks``: [(launches issued so far, lowest gradient offset written so far)] per backward layer (Plan.bwd_marks;
    the arena fills from its tail).  ``bounds``: ascending [(s, e)] buckets.  Returns [(end launch index, [(s, e), ...])]
    with strictly increasing ends, the last one == n_launches, every bucket exactly once; a bucket is attached to the
    first segment end at which ``lowest offset <= s``; the head bucket (and whatever the backward never reaches, e.g.
    a frozen encoder) goes with the last segment."""
    segs, pending = [], list(reversed(bounds))
    for end, low in marks:
        ready = []
        while len(pending) > 1 and pending[0][0] >= low:
            ready.append(pending.pop(0))
        if not ready:
            continue
        if segs and segs[-1][0] == end:
            segs[-1][1].extend(ready)
        else:
            segs.append((end, ready))
    if segs and segs[-1][0] == n_launches:
        segs[-1][1].extend(pending)
    else:
        segs.append((n_launches, pending))
    return segs


def two_phase_bounds(marks, numel, frac=0.7, align=4):
    """Two gradient ranges for the overlapped reducer: the arena's tail, final once ``frac`` of the backward launches
    have been issued (for U-Net/ResNet: decoder + stages 4-3, > 90 % of the bytes), is all-reduced under the rest of
    the backward (the high-resolution stage-1/2 and stem layers: few parameters, a third of the time); the small
    head follows after the last launch.  xGMI rings are per-link bound, so one large message beats many buckets, and
    every extra backward segment costs a graph launch (~0.07 ms measured) - hence two phases, not N buckets."""
    if not marks:
        return [(0, numel)]
    total = marks[-1][0]
    low = numel
    for end, lo in marks:
        low = lo
        if end >= frac * total:
            break
    low = (low // align) * align
    if low <= 0 or low >= numel:
        return [(0, numel)]
    return [(0, low), (low, numel)]


class _WireWork(object):
    """Work handle of a range reduced in the wire format: ``wait()`` = the collective's wait + the cast back into the fp32 arena."""

    def __init__(self, work, finish):
        self.work, self.finish = work, finish

    def wait(self):
        self.work.wait()
        self.finish()


class GradReducer(object):
    """Sum-all-reduce of a flat gradient arena in buckets; the mean's 1/world factor is NOT applied
    to the arena - read it from ``scale`` and fold it into the optimizer (HipSegModel.gscale)."""

    def __init__(self, group=None, bucket_mb=32.0, wire_bf16=False, cast_fns=None, force=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force          # run the collectives even with one rank (single-GPU test of the RCCL path)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.wire_bf16 = wire_bf16
        self.cast_fns = cast_fns        # (f32->bf16, bf16->f32) device kernels when wire_bf16
        self._wire = None
        self._bounds = None

    @property
    def scale(self):
        return 1.0 / self.world

    @property
    def active(self):
        return self.world > 1 or (self.force and dist.is_initialized())

    def bounds(self, numel):
        if self._bounds is None or self._bounds[-1][1] != numel:
            self._bounds = bucket_bounds(numel, self.bucket_elems)
        return self._bounds

    def set_bounds(self, bounds):
        """Replaces the uniform buckets (e.g. with two_phase_bounds for the overlapped schedule)."""
        self._bounds = list(bounds)

    def reset_bounds(self):
        """Back to the uniform buckets (the serialised schedule: one pass of `bucket_mb` collectives after the backward)."""
        self._bounds = None

    def allreduce_range(self, flat, s, e):
        """Asynchronous SUM-all-reduce of flat[s:e] ordered after the work already on the current stream; returns
        the work handle (``wait()`` orders the current stream after the collective) or None when inactive.  With the bf16 wire
        format the range is cast into the wire buffer first and cast back by ``wait()`` - half the bytes on xGMI, also on the
        overlapped path."""
        if not self.active or os.environ.get("STP_DP_NOCOMM") == "1":   # (NOCOMM: measure the segmentation cost alone)
            return None
        if self.wire_bf16:
            if self._wire is None or self._wire.numel() != flat.numel():
                self._wire = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            to_bf16, to_f32 = self.cast_fns
            to_bf16(flat[s:e], self._wire[s:e], e - s)
            return _WireWork(dist.all_reduce(self._wire[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                             lambda: to_f32(self._wire[s:e], flat[s:e], e - s))
        return dist.all_reduce(flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def allreduce(self, flat):
        if self.world <= 1 and not (self.force and dist.is_initialized()):
            return
        if self._bounds is None or self._bounds[-1][1] != flat.numel():
            self._bounds = bucket_bounds(flat.numel(), self.bucket_elems)
        if self.wire_bf16:
            if self._wire is None or self._wire.numel() != flat.numel():
                self._wire = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            to_bf16, to_f32 = self.cast_fns
            works = []
            for s, e in self._bounds:
                to_bf16(flat[s:e], self._wire[s:e], e - s)
                works.append(dist.all_reduce(self._wire[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for (s, e), w in zip(self._bounds, works):
                w.wait()
                to_f32(self._wire[s:e], flat[s:e], e - s)
            return
        # issued back to back on the current stream: RCCL pipelines consecutive collectives of one communicator
        for s, e in self._bounds:
            dist.all_reduce(flat[s:e], op=dist.ReduceOp.SUM, group=self.group)


def make_reducer(force=False):
    """The reducer of the training loops and bench.py: 32 MB buckets (STP_DP_BUCKET_MB), fp32 on the wire - the data-parallel step
    is then bit-identical to accumulating the ranks' gradients in one process.  STP_DP_WIRE=bf16 halves the bytes on xGMI (the
    97.7 MB fp32 arena of U-Net/ResNet34 becomes 48.9 MB, cast by stp_cast_f32_to_bf16 / stp_cast_bf16_to_f32 on the compute
    stream).  Error model of the bf16 wire: every rank's contribution is rounded to bf16 once, AND the collective itself sums in bf16 -
    a ring reduce-scatter rounds the running sum at each of its world-1 hops, so the relative error of a summed gradient grows from
    2^-8 (bf16's unit roundoff: one rounding) towards world x 2^-8 in the worst case (sqrt(world) x 2^-8 typically); replicas stay bit-identical to
    each other (all-gather copies one result).  tests/test_distributed_cpu.py::test_bf16_wire_error_model bounds it for two ranks;
    keep the fp32 wire where gradient accuracy matters more than the 0.2-0.3 ms of xGMI time."""
    wire = os.environ.get("STP_DP_WIRE", "fp32") == "bf16"
    cast = None
    if wire:
        from . import ops
        cast = (lambda s, d, c: ops.cast_f32_to_bf16(s, d, c), lambda s, d, c: ops.cast_bf16_to_f32(s, d, c))
    return GradReducer(bucket_mb=float(os.environ.get("STP_DP_BUCKET_MB", "32")), wire_bf16=wire, cast_fns=cast, force=force)


def active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def barrier():
    if active():
        dist.barrier()


def broadcast_tensors(tensors, src=0):
    """Rank ``src``'s values of every tensor replace the others' (parameters, BatchNormalization statistics, optimizer
    state after a checkpoint load: the replicas must start a stage bit-identical, nothing re-synchronises them later)."""
    if not active():
        return
    for t in tensors:
        if t is not None:
            dist.broadcast(t, src=src)


def average_tensor(t):
    """In-place mean over the ranks (BatchNormalization moving statistics at the end of an epoch)."""
    if active() and t is not None and t.numel():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.mul_(1.0 / dist.get_world_size())


def allreduce_sums(vec):
    """SUM over the ranks of a small float64 vector (epoch scalars: weighted metric sums + sample counts).  Every rank
    receives the same bits, so decisions derived from the result (best checkpoint, EarlyStopping, ReduceLROnPlateau) are
    identical on all ranks by construction.  Goes through a CPU tensor with gloo, a device tensor with RCCL."""
    import numpy as np
    v = np.asarray(vec, np.float64)
    if not active():
        return v
    t = torch.from_numpy(v.copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def replicas_equal(t):
    """True when every rank holds the same bits in ``t`` (exact integer checksum of the raw words, MAX-reduced with its
    negation): the data-parallel replicas must never drift - nothing re-synchronises them inside a stage."""
    if not active():
        return True
    words = t.detach().reshape(-1).view(torch.int32).to(torch.int64)
    c = int((words * (torch.arange(words.numel(), device=words.device, dtype=torch.int64) % 8191 + 1)).sum().item())
    v = torch.tensor([c, -c], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        v = v.cuda()
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    return int(v[0].item()) == -int(v[1].item())


def shard_list(indexes, rank, world):
    """Strided shard WITHOUT wrap-around (validation: no collective per batch, so the shards may differ by one sample;
    every sample is evaluated exactly once across the ranks)."""
    return list(indexes)[rank::world] if world > 1 else list(indexes)


def shard_indices(n, rank, world, epoch, seed):
    """Per-epoch permutation shared by all ranks (same seed), strided by rank; every rank gets
    the same number of samples (the tail wraps around) so collectives stay aligned."""
    g = torch.Generator()
    g.manual_seed(int(seed) * 1000003 + int(epoch))
    perm = torch.randperm(n, generator=g).tolist()
    per = (n + world - 1) // world
    perm = perm + perm[: per * world - n]
    return perm[rank::world][:per]
