"""Static training/inference plan: the host-side runtime of the HIP hot path.

A network definition (``nets.py``) is executed ONCE against a :class:`Plan`; every layer call
records the C-ABI launches of its forward, pushes a closure for its backward, and declares its
parameters inside flat fp32 arenas (master weights ``P``, gradients ``G``, optimizer state,
BN moving statistics ``S``).  The result is three launch lists - ``prep`` (weight compute
copies), ``fwd`` (+loss), ``bwd`` - plus ``opt``; a step replays them on one HIP stream, eagerly or
as a captured hipGraph.  There is no tracing compiler and no autograd: gradients meet by
plan-time bookkeeping (first writer overwrites, later writers accumulate in the GEMM epilogue).

PyTorch supplies device memory (``torch.empty``), the stream handle and - in ``distributed.py`` -
the RCCL process group; all arithmetic happens in libstp_hip.so.
"""
import os
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, ops

_TD = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


class StpShapeError(ValueError):
    pass


def _rup(a, b):
    return (a + b - 1) // b * b


class DT(object):
    """Device tensor handle: NHWC activation + (optional) gradient buffer."""
    __slots__ = ("name", "N", "H", "W", "C", "buf", "grad", "_grad_ready", "grad_writes", "needs_grad", "gradC", "meta")

    def __init__(self, name, N, H, W, Cn, buf=None, needs_grad=False):
        self.name, self.N, self.H, self.W, self.C = name, N, H, W, Cn
        self.buf, self.grad, self._grad_ready, self.needs_grad = buf, None, False, needs_grad
        self.grad_writes = 0
        self.gradC = Cn
        self.meta = {}

    @property
    def grad_ready(self):
        return self._grad_ready

    @grad_ready.setter
    def grad_ready(self, v):
        """Set by every consumer's backward after it wrote / accumulated its contribution: ``grad_writes`` counts them, so the
        consumer that finds ``grad_writes == uses - 1`` knows it completes the gradient (BatchNormalization-backward fusion)."""
        self._grad_ready = bool(v)
        if v:
            self.grad_writes += 1

    @property
    def rows(self):
        return self.N * self.H * self.W


class ParamInfo(object):
    __slots__ = ("name", "offset", "numel", "shape", "kind", "trainable", "pending_slices")

    def __init__(self, name, offset, shape, kind, trainable):
        self.name, self.offset, self.shape, self.kind, self.trainable = name, offset, tuple(shape), kind, trainable
        self.numel = int(np.prod(shape))
        self.pending_slices = 0       # column ranges of a shared 1x1 kernel whose weight gradient has not been planned yet (Plan.conv param_cols)


class Plan(object):
    def __init__(self, batch, dtype="bf16", device="cuda", training=True):
        if dtype not in _TD:
            raise ValueError("dtype must be 'bf16', 'fp16' or 'fp32'")
        self.device = torch.device(device)
        # A plan may be BUILT on CPU tensors (structure/shape checks in the CPU test suite);
        # it can only RUN on the GPU: run() raises otherwise - there is no CPU compute path.
        if self.device.type == "cuda" and not torch.cuda.is_available():
            raise _lib.StpError("the HIP training path needs a GPU; there is no CPU fallback")
        # the 16-bit storage format is a build parameter of the kernel set: fp16 plans run on libstp_hip_f16.so
        self.lib = _lib.load("fp16" if dtype == "fp16" else "bf16")
        self.loss_scale = 1.0          # fp16: the loss gradient is seeded times this; the optimizer's gscale divides it out
        self.dls = None                # fp16: device record of the dynamic multiplier on top of it (backend.HipSegModel, stp_scale_by_device)
        self.N = batch
        self.dtype = dtype
        self.tdt = _TD[dtype]
        self.cdt = {"bf16": ops.BF16, "fp16": ops.F16, "fp32": ops.F32}[dtype]
        self.vec = 4 if dtype == "fp32" else 8
        self.training = training
        self.params = OrderedDict()
        self.states = OrderedDict()
        self._poff = 0
        self._soff = 0
        self.dry = True
        self.frozen_prefixes = ()
        self._keep = []
        self.tensors = OrderedDict()
        self.P = self.G = self.S = None
        self.prep, self.fwd, self.bwd, self.opt = [], [], [], []
        self._tape = []
        self._goffs = []
        self.bwd_marks, self.bwd_monotone = [], True
        self._prep_layers = []
        self._upc_layers = []
        self._upc4_layers = []
        self._wg_ws_bytes = 0
        self._bn_ws_c = 4
        self.bn_momentum = 0.99
        self.fuse_bn_backward = os.environ.get("STP_FUSE_BN_BACKWARD", "1") != "0"
        self.fuse_bn_backward_last = os.environ.get("STP_FUSE_BN_BACKWARD_LAST", "1") != "0"
        # weight gradients (needed only by the optimizer / all-reduce) run on a second stream next to the data-gradient
        # + BatchNormalization-backward chain of the same layer: the small latency-bound kernels of one chain fill the
        # tails of the other's GEMMs (captured into the same hipGraph as a fork/join)
        # Round 3: OFF by default.  With the row-of-taps layers grouped into a few long launches (stp_wgrad_group_*) the second
        # stream no longer pays: same box, 40 graph steps, U-Net/ResNet34 bs16 - one stream 7.97 ms, two streams 8.07 ms (the grouped
        # launch holds every CU's LDS, the main chain waits for it either way), per-layer launches on two streams 8.38 ms
        # (profiles/r03c_ab.txt).  STP_SIDE_STREAM_WGRAD=1 restores the fork / join schedule.
        self.side_stream_wgrad = os.environ.get("STP_SIDE_STREAM_WGRAD", "0") != "0"
        self.fold_upsample_grad = os.environ.get("STP_FOLD_UPSAMPLE_GRAD", "1") != "0"
        self._side = None
        self._side_reads = set()
        # grouped weight gradients (stp_wgrad_group_*): the row-of-taps layers of a stage are collected and issued as ONE partial
        # + ONE reduce launch once their work reaches this many GFLOP (0: every layer is launched alone, as in round 2)
        self.wgrad_group_gflop = float(os.environ.get("STP_WGRAD_GROUP_GFLOP", "300"))
        # a LONE row-of-taps layer of the 128-channel class this large runs as a one-layer group on the all-taps kernel (_flush_wgroup):
        # FPN/ResNet50's `fpn_final` (512 -> 512 at 4 x 256 x 256, 1.24 TFLOP) 14.30 -> 14.04 and 14.62 -> 14.49 ms per step on two boxes
        # (profiles/r06e_step_ab.txt); 0 = never
        self.wgrad_lone_group_gflop = float(os.environ.get("STP_WGRAD_LONE_GROUP_GFLOP", "600"))
        # lone weight gradients with plain slabs (the bottleneck ResNets' 1x1 layers, stride-2 layers): their reduces can wait in a table and
        # run as ONE launch per STP_WGRAD_REDUCE_BATCH layers.  OPT-IN (default 0 = one reduce launch per layer): measured SLOWER on the
        # step - FPN/ResNet50 14.62 -> 14.77 ms, PSPNet/ResNet101 8.06 -> 8.07, U-Net/ResNet34 6.350 -> 6.370 (same box, two interleaved
        # repetitions, profiles/r06e_step_ab.txt): the 59 / 30 / 7 launches saved (~9 us each in the eager table, ~3 in the graph) cost less
        # than reading 1 - 16 MB of slabs per layer back from HBM instead of from the cache the partial launch just wrote them through
        self.reduce_batch = int(os.environ.get("STP_WGRAD_REDUCE_BATCH", "0")) if not self.side_stream_wgrad else 0
        self._pending_reduces, self._pending_reduce_hi = [], 0
        self._wgroup, self._wgroup_cls, self._wgroup_flops, self._wgroup_hi = [], 0, 0.0, 0
        self._wgroup_reads = set()      # dY buffers of the pending layers: nothing may rewrite them before the group is issued
        self.wgroups = []               # (layer names, class) of every issued group (inspection / tests)
        # STP_WGRAD_GROUP_JOIN=0: a grouped launch in flight is NOT joined at the next layer (only where a buffer it reads is
        # rewritten, and at the end of the launch list); per-layer weight-gradient chains keep their lag-1 join
        self._side_lag_join = os.environ.get("STP_WGRAD_GROUP_JOIN", "1") != "0"
        self._side_groups_only = True   # nothing but grouped launches has been forked since the last join
        self._dw_ws_bytes = 0
        self.step_state = None
        # fused BatchNormalization sums in fixed-point slots (stp_conv_params.stats_slots): no finalize kernels
        self.bn_slots = os.environ.get("STP_BN_SLOTS", "0") == "1"   # measured: no gain over the finalize kernels, kept opt-in
        # BatchNormalization whose only consumers are small-channel convolutions: normalised inside their halo staging
        # (stp_conv_params.src_bn_mean), the normalised tensor is never written
        self.fuse_bn_sc = os.environ.get("STP_FUSE_BN_SC", "1") != "0"
        # producer BatchNormalization applied in LDS by the halo kernel (needs its automatic selection: STP_HALO != 0) AND by the
        # weight gradient of the same layer (row-of-taps kernel, single-layer or grouped: conv_wgrad.hip's PBN instances): the
        # normalised tensor is never written, the stp_bn_apply launch disappears.  OPT-IN (STP_FUSE_BN_HALO=1): measured slower in
        # round 1 (the weight gradient still read the normalised tensor) and again in round 4 with the grouped weight gradient
        # normalising its operand itself (profiles/r04b_*): 6.99 vs 6.79 ms per step on one box.  The launches it removes cost
        # 178 + 45 us (31 stp_bn_apply / finalize_apply at 7-14 us, near the HBM roofline); the in-LDS transform costs the halo
        # forward +2.5..+6.5 us per launch (stage 1: +16 us: ~28 VALU instructions per 16-byte vector, by all 8 waves, with the MFMA
        # pipe idle) = +220 us, and the grouped weight gradients +140 us (every tile over a pixel range repeats the transform).
        self.fuse_bn_halo = os.environ.get("STP_FUSE_BN_HALO", "0") != "0" and os.environ.get("STP_HALO", "1") != "0"
        self.bn_slots_max_rows = int(os.environ.get("STP_BN_SLOTS_MAXROWS", "1073741824"))
        self._slot_need = 0          # int64 elements, counted in the dry pass
        self._slot_used = 0
        self.slot_arena = None
        self.loss_scalars = None
        self.inputs = {}

    # ------------------------------------------------------------------ definition driver
    def define(self, net_fn, share=None):
        """Runs ``net_fn(plan)`` twice: a dry pass that sizes the arenas, then the real pass.
        ``share``: another plan of the same network whose parameter / moving-statistics arenas
        are reused (the inference plan shares the training plan's weights)."""
        self.dry = True
        net_fn(self)
        n = _rup(self._poff, 4)
        if share is not None:
            if [(k, v.offset, v.shape) for k, v in share.params.items()] != [(k, v.offset, v.shape) for k, v in self.params.items()] \
                    or list(share.states.items()) != list(self.states.items()):
                raise StpShapeError("plans that share weights must declare identical parameters")
            self.P, self.S, self.G = share.P, share.S, None
        else:
            self.P = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.G = torch.zeros(n, dtype=torch.float32, device=self.device) if self.training else None
            self.S = torch.zeros(max(_rup(self._soff, 4), 4), dtype=torch.float32, device=self.device)
        self.mask = torch.ones(n, dtype=torch.uint8, device=self.device)
        self.ws_wgrad = torch.empty(max(self._wg_ws_bytes // 4, 4) + 4, dtype=torch.float32, device=self.device)
        self.ws_bn = torch.empty(ops.bn_workspace_bytes(_rup(self._bn_ws_c, 4)) // 4, dtype=torch.float32, device=self.device)
        self.ws_loss = torch.empty(ops.loss_workspace_bytes() // 4, dtype=torch.float32, device=self.device)
        self.ws_dw = torch.empty(max(self._dw_ws_bytes // 4, 4), dtype=torch.float32, device=self.device)
        if self.training and self.bn_slots and self._slot_need:
            self.slot_arena = torch.zeros(_rup(self._slot_need, 2), dtype=torch.int64, device=self.device)
        self.dry = False
        self._tape = []
        self._prep_layers = []
        self._upc_layers = []
        self._upc4_layers = []
        self.tensors = OrderedDict()
        net_fn(self)
        self._fuse_bn_into_consumers()
        self._fuse_bn_finalize()
        self._finish_prep()
        if self.training:
            # bwd_marks[i] = (launches issued after the i-th backward closure, lowest gradient offset written so far):
            # gradients at or above that offset are final from that launch on, PROVIDED layers finish in descending
            # arena order (true when parameters are declared in call order); otherwise bwd_monotone is cleared and
            # the reducer falls back to one all-reduce after the whole backward
            self.bwd_marks, self.bwd_monotone = [], True
            low = self.G.numel() if self.G is not None else 0
            for back in reversed(self._tape):
                self._goffs = []
                back()
                if self._goffs:
                    if max(e for _, e in self._goffs) > low:
                        self.bwd_monotone = False
                    low = min(low, min(o for o, _ in self._goffs))
                # a pending grouped weight gradient has asked for its addresses but not been launched: the arena is final only
                # above the highest pending layer
                # (... and below the highest layer whose reduce waits in the batched table)
                self.bwd_marks.append((len(self.bwd), max(low, self._wgroup_hi, self._pending_reduce_hi)))
            if self._wgroup or self._pending_reduces:
                self._flush_wgroup()
                self._flush_reduces()
                self.bwd_marks.append((len(self.bwd), low))
        self._tape = []
        return self

    @staticmethod
    def _nslots(Cn):
        """Slots per channel: as many as keep 2 * C * slots int64 reads in the consumer's prologue small (<= 16)."""
        n = 16
        while n > 1 and n * Cn > 1024:
            n //= 2
        return n

    def _slots(self, Cn):
        """(device pointer, slots per channel) of a zero-initialised [2][Cn][n] slice of the slot arena."""
        n = self._nslots(Cn)
        off = self._slot_used
        self._slot_used += 2 * Cn * n
        if self._slot_used > self.slot_arena.numel():
            raise StpShapeError("slot arena exhausted")
        return self.slot_arena.data_ptr() + 8 * off, n

    def _finish_prep(self):
        """One batched weight-preparation launch for all conv layers (descriptor table lives on the device)."""
        if self.slot_arena is not None:
            self._emit(self.prep, "stp_zero_bytes", self.slot_arena.data_ptr(), self.slot_arena.numel() * 8)
        n = len(self._prep_layers)
        if not n:
            return
        dsz = int(self.lib.stp_weight_prepare_desc_bytes())
        host = (C.c_char * (dsz * n))()
        total = 0
        for i, lay in enumerate(self._prep_layers):
            total += int(self.lib.stp_weight_prepare_desc_fill(C.cast(host, C.c_void_p), i, total, *lay))
        dev = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.device)
        self._keep.append(dev)
        self._emit(self.prep, "stp_weight_prepare_batched", dev.data_ptr(), n, total, self.cdt)
        if self._upc_layers:      # class-collapsed weight copies of the convolutions over upsample + concat (Plan.conv)
            import struct
            assert int(self.lib.stp_weight_prepare_upcollapse_desc_bytes()) == 32
            tab = b"".join(struct.pack("<QQiiii", *lay) for lay in self._upc_layers)
            udev = torch.frombuffer(bytearray(tab), dtype=torch.uint8).to(self.device)
            self._keep.append(udev)
            self._emit(self.prep, "stp_weight_prepare_upcollapse_batched", udev.data_ptr(), len(self._upc_layers), self.cdt)
            if self._upc4_layers:
                tab4 = b"".join(struct.pack("<QQiiii", *lay) for lay in self._upc4_layers)
                udev4 = torch.frombuffer(bytearray(tab4), dtype=torch.uint8).to(self.device)
                self._keep.append(udev4)
                self._emit(self.prep, "stp_weight_prepare_upcollapse_bwd_batched", udev4.data_ptr(), len(self._upc4_layers), self.cdt)

    # ------------------------------------------------------------------ parameters / state
    def param(self, name, shape, kind="weight"):
        if self.dry and name in self.params and tuple(self.params[name].shape) == tuple(shape) and self.params[name].kind == kind:
            return self.params[name]          # (a parameter several launches read: the column ranges of a shared 1x1 kernel, Plan.conv param_cols)
        if self.dry:
            off = _rup(self._poff, 4)
            trainable = not any(name.startswith(p) for p in self.frozen_prefixes)
            self.params[name] = ParamInfo(name, off, shape, kind, trainable)
            self._poff = off + int(np.prod(shape))
        return self.params[name]

    def state(self, name, numel, init=0.0):
        if self.dry:
            off = _rup(self._soff, 4)
            self.states[name] = (off, numel, init)
            self._soff = off + numel
        return self.states[name][0]

    def _pptr(self, info):
        return self.P.data_ptr() + 4 * info.offset

    def _gptr(self, info):
        # the backward closures ask for gradient addresses in backward order: the running minimum tells the
        # data-parallel reducer which tail of the arena is final after each layer (bwd_marks)
        self._goffs.append((info.offset, info.offset + int(np.prod(info.shape))))
        return self.G.data_ptr() + 4 * info.offset

    def _sptr(self, off):
        return self.S.data_ptr() + 4 * off

    # ------------------------------------------------------------------ buffers / launches
    def _alloc(self, shape, dtype=None):
        t = torch.empty(shape, dtype=dtype or self.tdt, device=self.device)
        self._keep.append(t)
        return t

    def _new(self, name, H, W, Cn, needs_grad, dtype=None):
        t = DT(name, self.N, H, W, Cn, None if self.dry else self._alloc((self.N, H, W, Cn), dtype), needs_grad)
        self.tensors[name] = t
        return t

    def _gradbuf(self, t):
        """Gradient buffer of ``t`` for a main-stream kernel that is about to WRITE it.  If a weight-gradient chain
        still in flight on the side stream reads that buffer (a dY aliased as a residual gradient), join first."""
        if t.grad is None:
            t.grad = self._alloc((t.N, t.H, t.W, t.gradC))
        if self._wgroup_reads and t.grad.data_ptr() in self._wgroup_reads:
            self._flush_wgroup()          # a pending grouped weight gradient reads this buffer as its dY: issue it first
        if self._side_reads and t.grad.data_ptr() in self._side_reads:
            self._mark(self.bwd, "join")
            self._side_reads.clear()
            self._side_groups_only = True
        return t.grad

    def _before_inplace_write(self, buf):
        """A main-stream kernel is about to rewrite ``buf`` in place (stp_relu_bwd masks a dY): a pending grouped weight gradient
        that still reads it as its dY is issued first, a side chain in flight that reads it is joined (the rule _gradbuf applies
        to gradient buffers it hands out; advisor finding, round 3: no current network aliases such a buffer, nothing guarded it)."""
        if buf is None:
            return
        if self._wgroup_reads and buf.data_ptr() in self._wgroup_reads:
            self._flush_wgroup()
        if self._side_reads and buf.data_ptr() in self._side_reads:
            self._mark(self.bwd, "join")
            self._side_reads.clear()
            self._side_groups_only = True

    @staticmethod
    def _use(*ts):
        """Counts the consumers of a tensor: a BatchNormalization output read by exactly one convolution gets its
        backward partial sums from that convolution's data-gradient epilogue (stp_conv_params.bnb_x)."""
        for t in ts:
            if t is not None:
                t.meta["uses"] = t.meta.get("uses", 0) + 1

    # a launch record is (C function, args without the trailing stream, entry-point name, meta);
    # meta carries the layer name and the ALGORITHMIC flops of GEMM launches for bench.py's roofline.
    def _emit(self, lst, fname, *args):
        lst.append((getattr(self.lib, fname), args, fname, None))

    def _fuse_bn_into_consumers(self):
        """BatchNormalization(+activation) outputs that only convolutions with a fused-producer path read:

        * every consumer is a small-channel convolution (conv_sc.hip: forward AND weight gradient normalise the pre-BN tensor
          while staging it): the stp_bn_apply launch is dropped; ``tensor(name)`` still materialises the tensor on demand;
        * some consumers are halo-kernel convolutions (conv_halo.hip: the forward normalises the slab in LDS): when every one of
          their weight gradients runs on the row-of-taps kernel (conv_wgrad.hip normalises its halo tile the same way) the launch
          is dropped as well; otherwise a weight gradient still reads the normalised tensor, the stp_bn_apply launch stays and
          moves to the SIDE stream - off the forward's critical chain conv -> finalize -> apply -> conv.

        The data gradient and the BatchNormalization backward never read the normalised tensor."""
        drop, side = set(), set()
        for t in self.tensors.values():
            rec, sc, halo = t.meta.get("apply_rec"), t.meta.get("sc_consumers") or [], t.meta.get("halo_consumers") or []
            if rec is None or not (sc or halo) or len(sc) + len(halo) != t.meta.get("uses", 0):
                continue
            pre, mean, rstd, gp, beta, relu = t.meta["bn"]
            for cp, wp in sc:
                cp.src0 = pre
                cp.src_bn_mean, cp.src_bn_rstd, cp.src_bn_gamma, cp.src_bn_beta, cp.src_bn_relu = mean, rstd, gp, beta, relu
                if not halo:       # (with halo consumers the normalised tensor exists anyway: the weight gradient reads it)
                    wp.src_bn_mean, wp.src_bn_rstd, wp.src_bn_gamma, wp.src_bn_beta, wp.src_bn_relu = mean, rstd, gp, beta, relu
            halo_full = bool(halo) and all(ok for _, _, ok in halo)      # every weight gradient fuses the BatchNormalization too
            for cp, wp, _ in halo:
                cp.src0 = pre
                cp.src_bn_mean, cp.src_bn_rstd, cp.src_bn_gamma, cp.src_bn_beta, cp.src_bn_relu = mean, rstd, gp, beta, relu
                if halo_full:
                    wp.src_bn_mean, wp.src_bn_rstd, wp.src_bn_gamma, wp.src_bn_beta, wp.src_bn_relu = mean, rstd, gp, beta, relu
            if halo_full:
                for cp, wp in sc:
                    wp.src_bn_mean, wp.src_bn_rstd, wp.src_bn_gamma, wp.src_bn_beta, wp.src_bn_relu = mean, rstd, gp, beta, relu
            if halo and not halo_full:
                side.add(id(rec))
            else:
                t.meta["deferred"] = rec
                t.meta["src_override"] = pre
                drop.add(id(rec))
        if drop or side:
            # every cross-stream edge of the step graph costs a queue hand-off: the deferred launches are issued in batches
            # behind ONE fork (their only readers run in the backward pass, so any point of the forward is early enough)
            batch = int(os.environ.get("STP_BN_SIDE_BATCH", "8"))
            out, pending = [], []

            def flush():
                if pending:
                    out.append((None, (), "fork", None))                      # the side stream waits for the statistics
                    out.extend(pending)
                    del pending[:]
            for r in self.fwd:
                if id(r) in drop:
                    continue
                if id(r) in side:
                    pending.append((r[0], r[1], r[2], dict(r[3] or {}, stream=1)))
                    if len(pending) >= batch:
                        flush()
                    continue
                out.append(r)
            flush()
            self.fwd = out

    def _fuse_bn_finalize(self):
        """stp_bn_finalize directly followed by the stp_bn_apply of the same BatchNormalization -> ONE stp_bn_finalize_apply launch where
        the library takes it (16-bit dtype, whole 64-channel slabs, <= 128 partial-sum columns): the apply pass reduces the partial
        sums of its own channel slab in its prologue, the single-workgroup-per-channel finalize launch (~5 us of latency on the
        critical chain conv -> finalize -> apply -> conv) disappears.  The backward pair is merged inside stp_bn_backward_fused(_add)."""
        if os.environ.get("STP_BN_FUSE_FINALIZE", "1") == "0":
            return
        out, i, fwd = [], 0, self.fwd
        while i < len(fwd):
            r = fwd[i]
            nxt = fwd[i + 1] if i + 1 < len(fwd) else None
            if (r[2] == "stp_bn_finalize" and nxt is not None and nxt[2] == "stp_bn_apply" and not (nxt[3] or {}).get("stream")
                    and nxt[1][7] == r[1][6] and nxt[1][8] == r[1][7] and nxt[1][1] == nxt[1][3] == self.cdt and nxt[1][5] == nxt[1][6] == r[1][3]
                    and nxt[1][12] == 0.0):
                part, tiles, rows, Cn, eps, mom, mean, rstd, mm, mv = r[1]
                x, _, y, _, rows2, _, _, _, _, gp, beta, relu, _ = nxt[1]
                if rows2 == rows and int(self.lib.stp_bn_finalize_apply_ok(self.cdt, rows, Cn, tiles)):
                    out.append((self.lib.stp_bn_finalize_apply, (part, tiles, x, y, self.cdt, rows, Cn, eps, mom, mean, rstd, mm, mv, gp, beta, relu),
                                "stp_bn_finalize_apply", None))
                    i += 2
                    continue
            out.append(r)
            i += 1
        self.fwd = out

    def tensor(self, name):
        """The named tensor with its buffer valid: a normalised tensor that only exists inside its consumers' staging is
        computed here with the launch the plan dropped (inspection / tests; not part of the step)."""
        t = self.tensors[name]
        rec = t.meta.get("deferred")
        if rec is not None:
            fn, args = rec[0], rec[1]
            rc = fn(*args, torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise _lib.StpError("%s failed with %d" % (rec[2], rc))
        return t

    def _scratch(self, nbytes):
        """(pointer, bytes) of a plan-time fp32 scratch buffer for a two-stage reduction; (None, 0) if the op needs none."""
        nbytes = int(nbytes)
        if nbytes <= 0:
            return None, 0
        buf = self._alloc((nbytes // 4,), torch.float32)
        return buf.data_ptr(), nbytes

    def _emit_side(self, lst, fname, *args):
        lst.append((getattr(self.lib, fname), args, fname, {"stream": 1}))

    @staticmethod
    def _mark(lst, what):
        lst.append((None, (), what, None))

    def _group_stats(self, p, st, Cs):
        """Group-level pre-reduction of a convolution's statistic columns (stp_conv_params.stats_group): where the kernel that serves
        ``p`` has the epilogue and its table has more than 128 columns, the launch also writes a [2][Cs][columns / G] table that the
        one-launch finalize + apply kernels take - returns (the table a BatchNormalization should read, its columns)."""
        cols = int(self.lib.stp_conv2d_stats_floats(C.byref(p))) // (2 * Cs)
        G = int(self.lib.stp_conv2d_stats_group_for(C.byref(p)))
        if G < 2:
            return st, cols
        ng = -(-cols // G)
        out = self._alloc((2 * Cs * ng,), torch.float32)
        cnt = torch.zeros((max(1, int(self.lib.stp_conv2d_stats_group_counters(C.byref(p), G))),), dtype=torch.int32, device=self.device)
        self._keep.append(cnt)
        p.stats_group_out, p.stats_group_counters, p.stats_group = out.data_ptr(), cnt.data_ptr(), G
        return out, ng

    def _emit_conv(self, lst, p, meta=None):
        self._keep.append(p)
        if meta is not None and meta.get("tile") == 512:
            meta = dict(meta, sc=(int(p.C0), 1 if p.Cout <= 16 else 2))    # the small-channel kernel's instance <Cin, 16-channel tiles> (bench.py)
        if meta is not None and meta.get("tile") == 800:
            meta = dict(meta, pw=(int(p.C0), int(p.Cout)))                  # the pointwise streaming kernel's instance <Cin, Cout, ...> (bench.py)
        lst.append((self.lib.stp_conv2d, (C.byref(p),), "stp_conv2d", meta))

    def _emit_wgrad(self, lst, p, meta=None, defer_hi=0):
        """Weight gradient = split partial sums + fixed-order reduce: two launch records so that each
        kernel can be timed on its own (bench.py) - same arithmetic as stp_conv2d_wgrad.
        ``defer_hi`` > 0 (the end of the layer's range in the gradient arena; only when nothing reads dW before the optimizer): a layer
        with plain slabs writes them into a workspace of its own and its reduce joins the pending table (_flush_reduces)."""
        self._keep.append(p)
        meta = dict(meta or {}, stream=1)
        if defer_hi > 0 and self.reduce_batch > 0 and lst is self.bwd:
            db = int(self.lib.stp_wgrad_reduce_desc_bytes())
            wsb = int(self.lib.stp_conv2d_wgrad_workspace_bytes(C.byref(p)))
            ws = self._alloc((max(wsb // 4, 4) + 4,), torch.float32)
            desc = (C.c_char * db)()
            count = int(self.lib.stp_wgrad_reduce_desc_fill(C.addressof(desc), 0, C.byref(p), ws.data_ptr()))
            if count > 0:
                lst.append((self.lib.stp_conv2d_wgrad_partial, (C.byref(p), ws.data_ptr(), ws.numel() * 4, 0), "stp_conv2d_wgrad", meta))
                self._pending_reduces.append((bytes(desc), count))
                self._pending_reduce_hi = max(self._pending_reduce_hi, int(defer_hi))
                if len(self._pending_reduces) >= self.reduce_batch:
                    self._flush_reduces()
                return
            self._keep = [t for t in self._keep if t is not ws]          # (row-of-taps slabs: the layer keeps its own reduce launch)
        lst.append((self.lib.stp_conv2d_wgrad_partial, (C.byref(p), self.ws_wgrad.data_ptr(), self.ws_wgrad.numel() * 4, 0),
                    "stp_conv2d_wgrad", meta))
        lst.append((self.lib.stp_conv2d_wgrad_reduce, (C.byref(p), self.ws_wgrad.data_ptr(), 0), "stp_conv2d_wgrad_reduce",
                    {"stream": 1}))

    def _flush_reduces(self):
        """One stp_wgrad_reduce_batched launch for the pending lone-layer reduces (see _emit_wgrad)."""
        if not self._pending_reduces:
            return
        table = b"".join(d for d, _ in self._pending_reduces)
        dev = torch.frombuffer(bytearray(table), dtype=torch.uint8).to(self.device)
        self._keep.append(dev)
        n, maxc = len(self._pending_reduces), max(c for _, c in self._pending_reduces)
        self.bwd.append((self.lib.stp_wgrad_reduce_batched, (dev.data_ptr(), n, maxc), "stp_wgrad_reduce_batched", {"stream": 1, "layers": n}))
        self._pending_reduces, self._pending_reduce_hi = [], 0

    def _flush_wgroup(self):
        """Issues the pending grouped weight gradient: descriptor table (built now - every pointer is final), one partial launch
        and one reduce launch on the side stream.  Each group owns its partial-slab workspace."""
        if not self._wgroup:
            return
        layers, cls = self._wgroup, self._wgroup_cls
        reads = self._wgroup_reads
        self._wgroup, self._wgroup_cls, self._wgroup_flops, self._wgroup_hi, self._wgroup_reads = [], 0, 0.0, 0, set()
        n = len(layers)
        # (... unless the lone layer is large enough to fill the all-taps kernel by itself: FPN's 512 -> 512 3x3 `fpn_final` at 4 x 256 x 256,
        #  1.24 TFLOP - 1284 us on the per-layer row-of-taps kernel at 963 TFLOP/s; STP_WGRAD_LONE_GROUP_GFLOP: the threshold, 0 = never)
        lone_group = cls == 128 and 0 < self.wgrad_lone_group_gflop * 1e9 <= layers[0][2]
        if n == 1 and cls != 32 and not lone_group:
            # a lone layer gains nothing from the work list (measured on the bottleneck ResNets, whose 3x3 layers never neighbour:
            # FPN/ResNet50 1024x1024 18.25 -> 17.85 ms, PSPNet/ResNet101 768x768 10.13 -> 10.03 ms with the per-layer launch and its
            # tuned split count); the 32-channel class exists only as a grouped kernel (the per-layer one pads it to 64)
            wp, name, flops = layers[0]
            self._mark(self.bwd, "fork")
            self._side_groups_only = False
            self._side_reads.update(reads)
            self._emit_wgrad(self.bwd, wp, {"layer": name, "pass": "wgrad", "flops": flops, "cout": wp.Cout,
                                            "sc": bool(self.lib.stp_wgrad_sc_eligible(C.byref(wp))),
                                            "kernel_id": int(self.lib.stp_conv2d_wgrad_kernel_id(C.byref(wp)))})
            return
        arr = (C.POINTER(_lib.WgradParams) * n)(*[C.pointer(wp) for wp, _, _ in layers])
        tb = int(self.lib.stp_wgrad_group_table_bytes(arr, n))
        wsb = int(self.lib.stp_wgrad_group_workspace_bytes(arr, n))
        if tb <= 0 or wsb <= 0:
            raise StpShapeError("grouped weight gradient: the layers %s do not form a group" % [nm for _, nm, _ in layers])
        host = (C.c_char * tb)()
        _lib.check(self.lib.stp_wgrad_group_build(arr, n, C.addressof(host), tb), "stp_wgrad_group_build")
        dev = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.device)
        ws = self._alloc((wsb // 4,), torch.float32)
        self._keep += [arr, host, dev] + [wp for wp, _, _ in layers]
        names = [nm for _, nm, _ in layers]
        self.wgroups.append((names, cls))
        self._mark(self.bwd, "fork")
        self._side_reads.update(reads)
        hdr = (C.c_int32 * 16).from_buffer_copy(bytes(host)[:64])      # WgGroupHeader: [14] = fused producer BN instance, [15] = all-taps tiles
        meta = {"layer": "group[%d]:%s..%s" % (n, names[0], names[-1]), "pass": "wgrad", "flops": sum(f for _, _, f in layers), "cout": cls,
                "bm": cls, "layers": names, "stream": 1, "taps9": int(hdr[15]), "pbn": int(hdr[14])}
        self.bwd.append((self.lib.stp_wgrad_group_partial, (C.addressof(host), dev.data_ptr(), ws.data_ptr(), wsb), "stp_wgrad_group_partial", meta))
        self.bwd.append((self.lib.stp_wgrad_group_reduce, (C.addressof(host), dev.data_ptr(), ws.data_ptr()), "stp_wgrad_group_reduce",
                         {"stream": 1}))

    # ------------------------------------------------------------------ layers
    def input_u8(self, name, H, W, Cn):
        t = DT(name, self.N, H, W, Cn, None if self.dry else self._alloc((self.N, H, W, Cn), torch.uint8))
        self.inputs[name] = t
        return t

    def input_bn(self, name, x, eps):
        """BatchNormalization(scale=False) on the raw uint8 image -> dtype tensor padded to 4 channels (8 for images of 4..7
        channels, reference segmentation.py:135-155 builds N-channel models) whose channel x.C is the constant 1 (see
        stp_stem_beta_grad) and whose remaining channels are also 1 (their stem weights are zero: the unpadding drops their gradient)."""
        if x.C > 7:
            raise StpShapeError("input_bn handles up to 7 image channels")
        Cp = 4 if x.C <= 3 else 8
        beta = self.param(name + "/beta", (x.C,), "beta")
        mm = self.state(name + "/moving_mean", x.C, 0.0)
        mv = self.state(name + "/moving_variance", x.C, 1.0)
        out = self._new(name, x.H, x.W, Cp, False)
        out.meta["real_c"] = x.C
        out.meta["input_bn_beta"] = beta
        if self.dry:
            return out
        if self.training:
            mean, rstd = self._alloc((x.C,), torch.float32), self._alloc((x.C,), torch.float32)
            self._emit(self.fwd, "stp_bn_stats", x.buf.data_ptr(), ops.U8, x.rows, x.C, eps, self.bn_momentum,
                       mean.data_ptr(), rstd.data_ptr(), self._sptr(mm), self._sptr(mv), self.ws_bn.data_ptr(),
                       self.ws_bn.numel() * 4)
            self._emit(self.fwd, "stp_bn_apply", x.buf.data_ptr(), ops.U8, out.buf.data_ptr(), self.cdt, x.rows, x.C, Cp,
                       mean.data_ptr(), rstd.data_ptr(), None, self._pptr(beta), 0, 1.0)
        else:
            self._emit(self.fwd, "stp_bn_inference", x.buf.data_ptr(), ops.U8, out.buf.data_ptr(), self.cdt, x.rows, x.C, Cp,
                       self._sptr(mm), self._sptr(mv), eps, None, self._pptr(beta), 0, 1.0)
        return out

    def input_cast(self, name, x):
        """Raw uint8 image -> dtype tensor padded to 4 channels (zeros), no normalisation: the input of the keras.applications
        VGG encoders, which segmentation_models feeds with raw pixels (no in-graph preprocessing)."""
        if x.C > 7:
            raise StpShapeError("input_cast handles up to 7 image channels")
        Cp = 4 if x.C <= 3 else 8
        out = self._new(name, x.H, x.W, Cp, False)
        out.meta["real_c"] = x.C
        if self.dry:
            return out
        zero, one = self._alloc((8,), torch.float32), self._alloc((8,), torch.float32)
        zero.zero_(); one.fill_(1.0)
        # y = x * 1 + 0 through the uint8 BatchNorm-apply kernel (mean 0, rstd 1, no gamma/beta), padded channel = 0
        self._emit(self.fwd, "stp_bn_apply", x.buf.data_ptr(), ops.U8, out.buf.data_ptr(), self.cdt, x.rows, x.C, Cp,
                   zero.data_ptr(), one.data_ptr(), None, None, 0, 0.0)
        return out

    def bn(self, name, x, eps, relu=True, scale=True, momentum=None):
        """``relu``: False/0 none, True/1 ReLU, 2 ReLU6 (MobileNetV2).  ``momentum``: Keras BatchNormalization momentum (0.99)."""
        Cn = x.C
        relu = int(relu)
        momentum = self.bn_momentum if momentum is None else float(momentum)
        gamma = self.param(name + "/gamma", (Cn,), "gamma") if scale else None
        beta = self.param(name + "/beta", (Cn,), "beta")
        mm = self.state(name + "/moving_mean", Cn, 0.0)
        mv = self.state(name + "/moving_variance", Cn, 1.0)
        trainable = beta.trainable
        out = self._new(name, x.H, x.W, Cn, x.needs_grad or trainable)
        self._bn_ws_c = max(self._bn_ws_c, Cn)
        self._use(x)
        if self.dry:
            self._slot_need += 4 * Cn * self._nslots(Cn)       # forward statistics + backward sums
            return out
        gp = self._pptr(gamma) if gamma else None
        if not self.training:
            self._emit(self.fwd, "stp_bn_inference", x.buf.data_ptr(), self.cdt, out.buf.data_ptr(), self.cdt, x.rows, Cn, Cn,
                       self._sptr(mm), self._sptr(mv), eps, gp, self._pptr(beta), int(relu), 0.0)
            return out
        mean, rstd = self._alloc((Cn,), torch.float32), self._alloc((Cn,), torch.float32)
        fused = x.meta.get("stats")
        slots = x.meta.get("stats_slots")
        if slots is not None:
            # statistics from the producing convolution's fixed-point slots: finalize + normalise + activation in one kernel
            self._emit(self.fwd, "stp_bn_apply_slots", x.buf.data_ptr(), out.buf.data_ptr(), self.cdt, x.rows, Cn, slots[0], slots[1], eps,
                       momentum, mean.data_ptr(), rstd.data_ptr(), self._sptr(mm), self._sptr(mv), gp, self._pptr(beta), int(relu))
        elif fused is not None:
            st, cp = fused
            tiles = int(self.lib.stp_conv2d_stats_floats(C.byref(cp))) // (2 * Cn)
            if x.meta.get("stats_table") is not None:
                st, tiles = x.meta["stats_table"]
            self._emit(self.fwd, "stp_bn_finalize", st.data_ptr(), tiles, x.rows, Cn, eps, momentum, mean.data_ptr(),
                       rstd.data_ptr(), self._sptr(mm), self._sptr(mv))
        else:
            self._emit(self.fwd, "stp_bn_stats", x.buf.data_ptr(), self.cdt, x.rows, Cn, eps, momentum, mean.data_ptr(),
                       rstd.data_ptr(), self._sptr(mm), self._sptr(mv), self.ws_bn.data_ptr(), self.ws_bn.numel() * 4)
        if slots is None:
            self._emit(self.fwd, "stp_bn_apply", x.buf.data_ptr(), self.cdt, out.buf.data_ptr(), self.cdt, x.rows, Cn, Cn,
                       mean.data_ptr(), rstd.data_ptr(), gp, self._pptr(beta), int(relu), 0.0)
        out.meta["bn"] = (x.buf.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gp, self._pptr(beta), int(relu))
        if self.fuse_bn_sc and slots is None:
            out.meta["apply_rec"] = self.fwd[-1]
            out.meta["sc_consumers"] = []
            out.meta["halo_consumers"] = []

        def back():
            if not out.needs_grad or not out.grad_ready:
                return
            # dx is always produced (it is cheap relative to skipping logic); frozen params are masked in the optimizer
            bslots = out.meta.get("bnb_slots")
            dadd = None
            if (x.needs_grad and x.grad_ready and x.grad is not None and x.grad.data_ptr() in self._wgroup_reads and bslots is None
                    and out.meta.get("bnb") is not None):
                # x's gradient so far is the dY of a convolution whose weight gradient waits in the pending group (the residual
                # branch aliases it): accumulate OUT OF PLACE - the sum lands in a fresh buffer, the dY stays intact
                dadd = x.grad
                x.grad = None
            dx = self._gradbuf(x) if x.needs_grad else self._alloc((x.N, x.H, x.W, Cn))
            if bslots is not None:
                self._emit(self.bwd, "stp_bn_backward_slots", x.buf.data_ptr(), out.grad.data_ptr(), dx.data_ptr(), self.cdt, x.rows, Cn,
                           mean.data_ptr(), rstd.data_ptr(), gp, bslots[0], bslots[1], self._gptr(gamma) if gamma else None,
                           self._gptr(beta), int(x.grad_ready and x.needs_grad))
                if x.needs_grad:
                    x.grad_ready = True
                return
            fused_b = out.meta.get("bnb")
            if fused_b is not None:
                # the only consumer's data-gradient epilogue already masked dY and reduced the per-tile sums
                st, q = fused_b
                tiles = q if isinstance(q, int) else int(self.lib.stp_conv2d_stats_floats(C.byref(q))) // (2 * Cn)
                if dadd is not None:
                    self._emit(self.bwd, "stp_bn_backward_fused_add", x.buf.data_ptr(), out.grad.data_ptr(), dx.data_ptr(), dadd.data_ptr(),
                               self.cdt, x.rows, Cn, mean.data_ptr(), rstd.data_ptr(), gp, st.data_ptr(), tiles,
                               self._gptr(gamma) if gamma else None, self._gptr(beta), 1, self.ws_bn.data_ptr(), self.ws_bn.numel() * 4)
                else:
                    self._emit(self.bwd, "stp_bn_backward_fused", x.buf.data_ptr(), out.grad.data_ptr(), dx.data_ptr(), self.cdt,
                               x.rows, Cn, mean.data_ptr(), rstd.data_ptr(), gp, st.data_ptr(), tiles,
                               self._gptr(gamma) if gamma else None, self._gptr(beta), int(x.grad_ready and x.needs_grad),
                               self.ws_bn.data_ptr(), self.ws_bn.numel() * 4)
                if x.needs_grad:
                    x.grad_ready = True
                return
            self._emit(self.bwd, "stp_bn_backward", x.buf.data_ptr(), out.grad.data_ptr(), dx.data_ptr(), self.cdt, x.rows, Cn,
                       mean.data_ptr(), rstd.data_ptr(), gp, self._pptr(beta), self._gptr(gamma) if gamma else None,
                       self._gptr(beta), int(relu), int(x.grad_ready and x.needs_grad), self.ws_bn.data_ptr(),
                       self.ws_bn.numel() * 4)
            if x.needs_grad:
                x.grad_ready = True

        self._tape.append(back)
        return out

    def conv3x3_taps(self, name, x, Cout, bias=False):
        """``Conv2D(Cout, 3x3, padding 1)`` with few output channels over many input channels (the class heads of FPN / PSPNet: 512 -> 3 /
        20) as a 1x1 convolution into 9 x Cout tap channels + stp_tapsum_fwd (bn_pool.hip: W_t . shift_t(x) = shift_t(W_t . x)).  The
        parameters keep the layer's names and shapes (``name/kernel`` = (Cout, 3, 3, Cin), ``name/bias``): the 1x1 launch reads the 3x3
        kernel's own bytes as [9 Cout][Cin], its weight gradient IS the kernel's gradient."""
        cin = x.meta.get("real_c", x.C)
        # (the gradient of the tap channels in rows of 64: the data gradient's K is then whole 64-channel steps of the buffer-DMA kernel -
        #  with K = 32 it ran at 1.8 TB/s, 302 us on FPN's 4 x 256 x 256 x 512 map)
        z = self.conv(name + "_taps", x, 9 * Cout, 1, param_name=name, param_shape=(Cout, 3, 3, cin), cout_pad=64 if self.dtype != "fp32" else None)
        b = self.param(name + "/bias", (Cout,), "bias") if bias else None
        out = self._new(name, x.H, x.W, Cout, z.needs_grad or (b is not None and b.trainable))
        out.gradC = _rup(Cout, self.vec)
        if bias:
            self._bn_ws_c = max(self._bn_ws_c, out.gradC)
        self._use(z)
        if self.dry:
            return out
        self._emit(self.fwd, "stp_tapsum_fwd", z.buf.data_ptr(), out.buf.data_ptr(), self._pptr(b) if b is not None else None, self.N, x.H, x.W,
                   Cout, z.C, Cout, self.cdt)
        if not self.training:
            return out

        def back():
            if not out.grad_ready:
                return
            dy = out.grad
            if b is not None and b.trainable:
                tmp = self._alloc((out.gradC,), torch.float32)
                self._emit(self.bwd, "stp_channel_sum", dy.data_ptr(), self.cdt, out.rows, out.gradC, tmp.data_ptr(), 0,
                           self.ws_bn.data_ptr(), self.ws_bn.numel() * 4)
                self._emit(self.bwd, "stp_weight_grad_unpad", tmp.data_ptr(), self._gptr(b), Cout, 1, 1, 1, 1, 1, 0)
            if z.needs_grad:
                self._emit(self.bwd, "stp_tapsum_bwd", dy.data_ptr(), self._gradbuf(z).data_ptr(), self.N, x.H, x.W, Cout, out.gradC, z.gradC, self.cdt)
                z.grad_ready = True

        self._tape.append(back)
        return out

    def conv(self, name, x, Cout, k, stride=1, pad=0, src1=None, upsample=False, bias=False, residual=None, bn_stats=False,
             transpose=False, relu=False, same_tf=False, fold_shortcut=None, param_name=None, param_shape=None, cout_pad=None,
             param_cols=None, flops_as=None):
        """Conv2D (explicit symmetric ZeroPadding + 'valid').  ``upsample`` folds UpSampling2D(2) of x,
        ``src1`` folds Concatenate([up(x), src1]) into the GEMM gather; ``residual`` folds Add().

        ``fold_shortcut``: the output tensor of the 1x1 / stride-2 projection shortcut that reads the same ``x`` as this 3x3 / stride-2
        convolution (ResNet basic block): its data gradient is folded into this layer's data-gradient launch (stp_conv_params.fold_*).

        ``param_cols`` = (first column, total columns): this 1x1 convolution multiplies by the COLUMN RANGE [first, first + Cin) of the shared
        kernel ``param_name/kernel`` of shape (Cout, 1, 1, total) - the PSPNet head without its concatenation (nets.pspnet_resnet): the range is
        copied into a dense fp32 matrix in front of the weight preparation (stp_copy_cols_f32) and the weight gradient is copied back into it.

        ``transpose``: Keras ``Conv2DTranspose(Cout, k, strides=2, padding='same')`` (k even, 4 in segmentation_models'
        transpose decoder blocks).  TF pads the equivalent forward convolution by k/2-1 on each side, so the transposed
        convolution is a stride-1 convolution over the ZERO-INSERTED input (2H-1 x 2W-1) with pad k-1-(k/2-1) and the
        spatially flipped kernel - the zero-insertion gather that the stride-2 data-gradients already use.  The master
        parameter holds the flipped OHWI kernel (kind "tkernel"; set/get_weights convert from Keras' (kh,kw,out,in))."""
        if transpose:
            if stride != 1 or upsample or src1 is not None or residual is not None or k % 2:
                raise StpShapeError("transpose=True is the plain stride-2 'same' Conv2DTranspose with an even kernel")
            pad = k - 1 - (k // 2 - 1)
        real_c0 = x.meta.get("real_c", x.C)
        stem = real_c0 != x.C
        if stem and (src1 is not None or upsample):
            raise StpShapeError("padded-channel input supports a plain conv only")
        C0, C1 = x.C, (src1.C if src1 is not None else 0)
        if not stem and (C0 % self.vec or C1 % self.vec):
            raise StpShapeError("%s: input channels (%d,%d) must be multiples of %d for dtype %s" % (name, C0, C1, self.vec, self.dtype))
        Hv, Wv = (2 * x.H, 2 * x.W) if upsample else ((2 * x.H - 1, 2 * x.W - 1) if transpose else (x.H, x.W))
        if src1 is not None and (src1.H, src1.W) != (Hv, Wv):
            raise StpShapeError("%s: skip tensor is %dx%d, expected %dx%d" % (name, src1.H, src1.W, Hv, Wv))
        Ho, Wo = (Hv + 2 * pad - k) // stride + 1, (Wv + 2 * pad - k) // stride + 1
        if same_tf:
            # TF / Keras padding='same': ceil(size / stride) outputs, the odd padding pixel goes to the bottom / right.  The
            # kernels take the top/left padding and the output size; taps past the far edge are out of bounds = zero.
            Ho, Wo = -(-Hv // stride), -(-Wv // stride)
            pad = max((Ho - 1) * stride + k - Hv, 0) // 2
            if max((Wo - 1) * stride + k - Wv, 0) // 2 != pad:
                raise StpShapeError("%s: 'same' padding differs between height and width" % name)
        KWp = k + (k & 1) if (stem and x.C == 4) else k      # 4 padded channels: one 16-byte vector = two horizontally adjacent taps
        Cin_master = real_c0 + C1
        Cinp = C0 + C1
        # (param_name / param_shape: conv3x3_taps - the 1x1 launch over the bytes of a 3x3 kernel registered under the layer's own name)
        if param_cols is not None:
            if k != 1 or stem or transpose or C1 or Cout % self.vec or Cin_master % 4 or param_cols[0] % 4 or param_cols[1] % 4 \
                    or param_cols[0] + Cin_master > param_cols[1]:
                raise StpShapeError("%s: a column range of a shared kernel serves a plain 1x1 convolution with aligned channel counts" % name)
            param_shape = None
        w = self.param((param_name or name) + "/kernel", (Cout, 1, 1, param_cols[1]) if param_cols is not None else (param_shape or (Cout, k, k, Cin_master)),
                       "tkernel" if transpose else "kernel")
        if param_shape is not None and int(np.prod(param_shape)) != Cout * k * k * Cin_master:
            raise StpShapeError("%s: parameter view of %s does not match %d x %d x %d x %d" % (name, param_shape, Cout, k, k, Cin_master))
        b = self.param(name + "/bias", (Cout,), "bias") if bias else None
        CoutB = _rup(Cout, cout_pad or self.vec)      # channels of the gradient buffer = K of the data gradient (cout_pad: conv3x3_taps)
        x_ng = x.needs_grad
        s_ng = src1.needs_grad if src1 is not None else False
        out = self._new(name, Ho, Wo, Cout, x_ng or s_ng or w.trainable or (residual is not None and residual.needs_grad))
        out.gradC = CoutB
        if bias:
            self._bn_ws_c = max(self._bn_ws_c, CoutB)      # the bias gradient (stp_channel_sum) shares the BN workspace
        self._use(x, src1, residual)
        # workspace sizing needs the wgrad plan: query the library (cheap, host only)
        wp = _lib.WgradParams()
        wp.N, wp.Hs0, wp.Ws0, wp.Hv, wp.Wv, wp.C0, wp.C1 = self.N, x.H, x.W, Hv, Wv, C0, C1
        src_mode = ops.SRC_NEAREST2X if upsample else (ops.SRC_ZEROINS2X if transpose else ops.SRC_DIRECT)
        wp.src0_mode = src_mode
        wp.KH, wp.KW, wp.stride, wp.pad, wp.Ho, wp.Wo, wp.Cout = k, KWp, stride, pad, Ho, Wo, CoutB
        wp.accumulate, wp.dtype, wp.splits = 0, self.cdt, 0
        if self.training and w.trainable:
            self._wg_ws_bytes = max(self._wg_ws_bytes, int(self.lib.stp_conv2d_wgrad_workspace_bytes(C.byref(wp))))
        if self.dry:
            return out
        rows_f, rows_b = _rup(Cout, 16), _rup(Cinp, 16)
        wf = self._alloc((rows_f * k * KWp * Cinp,))
        need_dgrad = self.training and (x_ng or s_ng) and not stem
        wb = self._alloc((rows_b * k * k * CoutB,)) if need_dgrad else None
        out.meta["wb"] = wb
        wsrc = self._pptr(w)
        if param_cols is not None:
            w.pending_slices += 1      # (backward: the LAST range written makes the parameter's gradient final)
            # the column range as a dense [Cout][Cin] fp32 matrix, refreshed every step in front of the batched weight preparation
            wm = self._alloc((Cout * Cin_master,), torch.float32)
            self._emit(self.prep, "stp_copy_cols_f32", wm.data_ptr(), Cin_master, wsrc + 4 * int(param_cols[0]), int(param_cols[1]), Cout, Cin_master, 0)
            wsrc = wm.data_ptr()
        out.meta["w_master"] = (wsrc, Cout, Cin_master, k)      # (the space-to-depth data gradient builds its weights from the masters)
        # collected here, issued as ONE batched launch per step (see _finish_prep)
        self._prep_layers.append((wsrc, wf.data_ptr(), wb.data_ptr() if wb is not None else None,
                                  Cout, k, k, Cin_master, KWp, Cinp, CoutB))
        p = ops.conv_params(x.buf, wf, out.buf, N=self.N, Hs0=x.H, Ws0=x.W, Hv=Hv, Wv=Wv, C0=C0, C1=C1,
                            src1=src1.buf if src1 is not None else None,
                            mode=src_mode, KH=k, KW=KWp, stride=stride, pad=pad,
                            Ho=Ho, Wo=Wo, Cout=Cout, dtype=self.cdt, residual=residual.buf if residual is not None else None)
        w4 = None
        if (upsample and src1 is not None and k == 3 and KWp == 3 and stride == 1 and pad == 1 and Cinp == Cin_master == C0 + C1
                and os.environ.get("STP_UPCOLLAPSE", "1") != "0" and not self.lib.stp_conv2d_scn_eligible(C.byref(p))
                and int(self.lib.stp_conv2d_halo_variant(C.byref(p))) < 0):
            # (the narrow-output kernel - 64 + 64 -> 32 channels - keeps both halos in LDS and takes the plain weight copy; so does the
            #  two-source form of the halo kernel, round 5: 128+ output channels)
            # decoder conv1 = conv3x3(concat(UpSampling2D(2)(x), skip)): per output parity class the taps over the upsampled half read
            # 2 x 2 low-resolution pixels - the forward multiplies them by class-summed weights (4 x C0 + 9 x C1 K columns instead
            # of 9 x (C0 + C1)); the summed copy is rebuilt from the fp32 master with the other weight copies, once per step
            wup = self._alloc((rows_f * 16 * C0,))
            self._upc_layers.append((self._pptr(w), wup.data_ptr(), Cout, rows_f, C0, C0 + C1))     # one batched launch (_finish_prep)
            p.weight_up = wup.data_ptr()
            if need_dgrad and x_ng and C0 % 16 == 0 and os.environ.get("STP_UPCOLLAPSE_BWD", "0") == "1":
                # ... and the data gradient w.r.t. x is a 4x4 / stride-2 convolution of dY with the row / column tap sums.  OPT-IN:
                # measured slower with today's kernels for the two resulting shapes, kept for the next round
                w4 = self._alloc((C0 * 16 * CoutB,))
                self._upc4_layers.append((self._pptr(w), w4.data_ptr(), Cout, CoutB, C0, C0 + C1))
        if (self.training and x.meta.get("apply_rec") is not None and src1 is None and residual is None and not transpose and not stem
                and self.lib.stp_conv2d_sc_eligible(C.byref(p)) and (not w.trainable or self.lib.stp_wgrad_sc_eligible(C.byref(wp)))):
            x.meta["sc_consumers"].append((p, wp))       # see _fuse_bn_into_consumers
        elif (self.training and self.fuse_bn_halo and x.meta.get("apply_rec") is not None and src1 is None and not transpose and not stem
                and not upsample and self.lib.stp_conv2d_halo_variant(C.byref(p)) >= 0):
            # (p, wp, can the weight gradient normalise its operand itself: row-of-taps kernel - or no weight gradient at all)
            x.meta["halo_consumers"].append((p, wp, (not w.trainable) or int(self.lib.stp_conv2d_wgrad_kernel_id(C.byref(wp))) in (2, 3)))
        if b is not None:
            p.bias = self._pptr(b)
        if relu:
            if CoutB != Cout:
                raise StpShapeError("%s: a fused ReLU needs Cout to be a multiple of %d" % (name, self.vec))
            p.relu = 1        # Conv2D(activation='relu'): fused into the epilogue; its gradient masks dY first (stp_relu_bwd)
        if bn_stats and self.training and self.slot_arena is not None and self.N * Ho * Wo <= self.bn_slots_max_rows:
            sp, sn = self._slots(Cout)
            p.stats_partial, p.stats_slots = sp, sn
            out.meta["stats_slots"] = (sp, sn)
        elif bn_stats and self.training:
            # the BatchNormalization that follows takes its batch statistics from this conv's epilogue
            nfl = int(self.lib.stp_conv2d_stats_floats(C.byref(p)))
            st = self._alloc((max(nfl, 4),), torch.float32)
            p.stats_partial = st.data_ptr()
            out.meta["stats"] = (st, p)
            out.meta["stats_table"] = self._group_stats(p, st, Cout)       # (table, columns) the BatchNormalization reads
        # algorithmic work of this layer: 2 * pixels * Cout * KH*KW*Cin with the REAL (unpadded) dims
        flops = 2.0 * self.N * Ho * Wo * Cout * k * k * Cin_master / (4.0 if transpose else 1.0)   # zero-inserted taps are not work
        if flops_as is not None:
            # (a launch of a RESTRUCTURED reference layer - the PSPNet head without its concatenation: the roofline bookkeeping keeps
            #  counting the reference graph's convolution, SURVEY 8d's convention, not the cheaper form that is executed)
            flops = float(flops_as)
        self._emit_conv(self.fwd, p, {"layer": name, "pass": "fwd", "flops": flops, "tile": int(self.lib.stp_conv2d_tile_for(C.byref(p))),
                                      "src2": bool(C1 or upsample)})
        if not self.training:
            return out

        def back():
            if not out.needs_grad or not out.grad_ready:
                return
            dy = out.grad
            rows = out.rows
            if relu:
                self._before_inplace_write(dy)
                self._emit(self.bwd, "stp_relu_bwd", out.buf.data_ptr(), dy.data_ptr(), rows * out.gradC, self.cdt)
            # lag-1 join: the previous convolution's weight-gradient chain finishes before this layer's kernels start.
            # (Letting the side chain fall further behind - joining only on a buffer hazard, see _gradbuf - measured
            # SLOWER, 11.15 vs 10.88 ms/step: the chain then reads dY / x long after the main chain left them in L2.)
            if self._side_lag_join or not self._side_groups_only:
                self._mark(self.bwd, "join")
                self._side_reads.clear()
                self._side_groups_only = True
            # residual branch: d(residual) = dY
            if residual is not None and residual.needs_grad:
                if not residual.grad_ready and residual.gradC == out.gradC:
                    residual.grad = dy          # alias: dY is dead once this layer's backward has been issued
                    residual.grad_ready = True
                else:
                    self._emit(self.bwd, "stp_add_inplace", self._gradbuf(residual).data_ptr(), dy.data_ptr(),
                               rows * out.gradC, self.cdt)
            # weight gradient: on the side stream, forked here (dY is final); joined before any kernel rewrites dY (_gradbuf)
            # and at the end of the launch list
            if w.trainable:
                padded = stem or CoutB != Cout
                gcols = None
                if padded:
                    dwp = self._alloc((CoutB * k * KWp * Cinp,), torch.float32)
                    wp.dw = dwp.data_ptr()
                elif param_cols is not None:
                    gcols = self._alloc((Cout * Cin_master,), torch.float32)       # dense gradient of the column range, copied back below
                    wp.dw = gcols.data_ptr()
                else:
                    wp.dw = self._gptr(w)
                wp.src0, wp.src1, wp.dy = x.meta.get("src_override") or x.buf.data_ptr(), (src1.buf.data_ptr() if src1 is not None else None), dy.data_ptr()
                cls = int(self.lib.stp_wgrad_group_class(C.byref(wp))) if (self.wgrad_group_gflop > 0 and not padded and gcols is None) else 0
                if cls:
                    # row-of-taps layer: joins the pending group (one launch per stage instead of one per layer); dY stays untouched
                    # until the group is issued (_gradbuf / the BatchNormalization backward's out-of-place accumulate see to that)
                    if self._wgroup and self._wgroup_cls != cls:
                        self._flush_wgroup()
                    self._wgroup.append((wp, name, flops))
                    self._wgroup_cls = cls
                    self._wgroup_flops += flops
                    self._wgroup_hi = max(self._wgroup_hi, w.offset + int(np.prod(w.shape)))
                    self._wgroup_reads.add(dy.data_ptr())
                    if self._wgroup_flops >= self.wgrad_group_gflop * 1e9:
                        self._flush_wgroup()
                else:
                    # a layer outside the groups (stride 2, 1x1, stem, small-channel): the pending group is issued first, so a group
                    # = consecutive row-of-taps layers (a network stage) and the gradient arena stays final above the last visited layer
                    self._flush_wgroup()
                    self._mark(self.bwd, "fork")
                    self._side_groups_only = False
                    self._side_reads.add(dy.data_ptr())     # see _gradbuf: the only buffer of the chain that is ever rewritten
                    self._emit_wgrad(self.bwd, wp, {"layer": name, "pass": "wgrad", "flops": flops, "cout": CoutB,
                                                    "sc": bool(self.lib.stp_wgrad_sc_eligible(C.byref(wp))),
                                                    "kernel_id": int(self.lib.stp_conv2d_wgrad_kernel_id(C.byref(wp)))},
                                     defer_hi=0 if (padded or gcols is not None) else w.offset + int(np.prod(w.shape)))       # (a padded dW is unpadded right below)
                if gcols is not None:
                    # (the arena range of the shared kernel is reported final - _gptr, bwd_marks - by the last of its ranges only)
                    w.pending_slices -= 1
                    gbase = self._gptr(w) if w.pending_slices == 0 else self.G.data_ptr() + 4 * w.offset
                    self._emit_side(self.bwd, "stp_copy_cols_f32", gbase + 4 * int(param_cols[0]), int(param_cols[1]), gcols.data_ptr(), Cin_master,
                                    Cout, Cin_master, 0)
                if padded:
                    self._emit_side(self.bwd, "stp_weight_grad_unpad", dwp.data_ptr(), self._gptr(w), Cout, k, k, Cin_master, KWp,
                                    Cinp, 0)
                beta = x.meta.get("input_bn_beta")
                if stem and beta is not None and beta.trainable:
                    self._emit_side(self.bwd, "stp_stem_beta_grad", dwp.data_ptr(), self._pptr(w), self._gptr(beta), Cout, k, k,
                               real_c0, KWp, Cinp, real_c0)
            if b is not None and b.trainable and out.meta.get("loss_bias_grad") and Cout == 1:
                # the 1-class head: the loss gradient kernel left the per-workgroup sums of dL/dlogit in its workspace
                self._emit(self.bwd, "stp_sigmoid_loss_bias_grad", self.ws_loss.data_ptr(), rows, self._gptr(b), 0)
            elif b is not None and b.trainable:
                tmp = self._alloc((CoutB,), torch.float32)
                self._emit(self.bwd, "stp_channel_sum", dy.data_ptr(), self.cdt, rows, CoutB, tmp.data_ptr(), 0,
                           self.ws_bn.data_ptr(), self.ws_bn.numel() * 4)
                self._emit(self.bwd, "stp_weight_grad_unpad", tmp.data_ptr(), self._gptr(b), Cout, 1, 1, 1, 1, 1, 0)
            # data gradient
            if need_dgrad and out.meta.get("dgrad_folded"):
                pass        # a projection shortcut whose data gradient rode in its sibling's launch (fold_shortcut): nothing to issue
            elif need_dgrad:
                d0_hires = None
                if upsample:
                    d0 = d0_hires = self._alloc((self.N, Hv, Wv, C0)) if x_ng else None      # released below when the launch folds the 2 x 2 sums
                    acc0 = 0
                else:
                    d0 = self._gradbuf(x) if x_ng else None
                    acc0 = int(x.grad_ready)
                d1 = self._gradbuf(src1) if s_ng else None
                acc1 = int(src1.grad_ready) if s_ng else 0
                if d0 is None:
                    d0 = self._alloc((self.N, Hv, Wv, C0))      # gradient not wanted: scratch sink
                if C1 and d1 is None:
                    d1 = self._alloc((self.N, Hv, Wv, C1))
                if transpose:
                    # gradient of the zero-inserted input at its even positions only = a plain stride-2 convolution of dY
                    q = ops.conv_params(dy, wb, d0, N=self.N, Hs0=Ho, Ws0=Wo, Hv=Ho, Wv=Wo, C0=CoutB, mode=ops.SRC_DIRECT, KH=k, KW=k,
                                        stride=2, pad=k - 1 - pad, Ho=x.H, Wo=x.W, Cout=C0, dtype=self.cdt, accumulate0=acc0)
                else:
                    q = ops.conv_params(dy, wb, d0, N=self.N, Hs0=Ho, Ws0=Wo,
                                        Hv=(2 * Ho - 1 if stride == 2 else Ho), Wv=(2 * Wo - 1 if stride == 2 else Wo),
                                        C0=CoutB, mode=(ops.SRC_ZEROINS2X if stride == 2 else ops.SRC_DIRECT), KH=k, KW=k, stride=1,
                                        pad=k - 1 - pad, Ho=Hv, Wo=Wv, Cout=C0 + C1, dtype=self.cdt, dst1=d1, Cd0=C0,
                                        accumulate0=acc0, accumulate1=acc1)
                if stride not in (1, 2):
                    raise StpShapeError("data gradient supports stride 1 and 2")
                if (stride == 2 and k == 1 and pad == 0 and not transpose and not upsample and src1 is None and x_ng and fold_shortcut is None
                        and C0 % 4 == 0 and os.environ.get("STP_SCATTER_1X1S2", "1") != "0"):
                    # 1x1 / stride 2 (the projection shortcut of a bottleneck ResNet's first unit): the zero-inserted form above runs the GEMM
                    # over all four parity classes of the high-resolution grid (229 us for 256 <- 512 channels at 4 x 256 x 256).  Instead:
                    # t = W^T dY at LOW resolution (a plain 1x1 / stride-1 launch), then one pass that puts t at the even positions of the
                    # gradient - and, when that completes the gradient of a BatchNormalization output, masks it and reduces the sums
                    t_low = self._alloc((self.N, Ho, Wo, C0))
                    qg = ops.conv_params(dy, wb, t_low, N=self.N, Hs0=Ho, Ws0=Wo, Hv=Ho, Wv=Wo, C0=CoutB, mode=ops.SRC_DIRECT, KH=1, KW=1,
                                         stride=1, pad=0, Ho=Ho, Wo=Wo, Cout=C0, dtype=self.cdt)
                    self._emit_conv(self.bwd, qg, {"layer": name, "pass": "dgrad", "flops": 2.0 * self.N * Ho * Wo * Cout * Cin_master,
                                                   "tile": int(self.lib.stp_conv2d_tile_for(C.byref(qg)))})
                    uses, bnm = x.meta.get("uses", 0), x.meta.get("bn")
                    done = (uses == 1 and not acc0) or (self.fuse_bn_backward_last and uses > 1 and x.grad_writes == uses - 1 and acc0)
                    ntl = int(self.lib.stp_scatter2x_bwd_bn_tiles(self.N, x.H, x.W, C0, self.cdt)) if (
                        self.fuse_bn_backward and bnm is not None and done and self.slot_arena is None) else 0
                    if ntl > 0:
                        st = self._alloc((2 * C0 * ntl,), torch.float32)
                        self._emit(self.bwd, "stp_scatter2x_bwd_bn", t_low.data_ptr(), d0.data_ptr(), self.N, x.H, x.W, C0, self.cdt, acc0,
                                   bnm[0], bnm[1], bnm[2], bnm[3], bnm[4], bnm[5], st.data_ptr())
                        x.meta["bnb"] = (st, ntl)
                    else:
                        self._emit(self.bwd, "stp_scatter2x_bwd", t_low.data_ptr(), d0.data_ptr(), self.N, x.H, x.W, C0, self.cdt, acc0)
                    x.grad_ready = True
                    return
                fs = fold_shortcut
                s2d = False
                if (stride == 2 and k == 3 and pad == 1 and not transpose and src1 is None and x_ng and self.dtype != "fp32" and not stem
                        and Cout == CoutB and C0 == Cin_master and (Hv, Wv) == (2 * Ho, 2 * Wo) and os.environ.get("STP_S2D", "1") != "0"
                        and os.environ.get("STP_HALO", "1") != "0"):      # (stp_conv2d_halo_variant ignores the A/B switch; the dispatcher honours it)
                    # SPACE-TO-DEPTH form (round 5, stp_conv_params.s2d_dgrad): the four output parity classes as ONE dense 2 x 2-tap
                    # convolution of dY into 4 x C0 class-major channels on the halo kernel, stored depth-to-space; the sibling 1x1 /
                    # stride-2 shortcut's dY rides along as a second source (its weights live at class 0 / tap 0 only)
                    fold = (fs is not None and fs.needs_grad and fs.grad_ready and fs.meta.get("wb") is not None and fs.gradC == CoutB
                            and (fs.H, fs.W) == (Ho, Wo) and fs.meta.get("w_master", (0, 0, 0, 0))[1:] == (Cout, C0, 1))
                    qs = ops.conv_params(dy, dy, d0, N=self.N, Hs0=Ho, Ws0=Wo, Hv=Ho, Wv=Wo, C0=CoutB, C1=(CoutB if fold else 0),
                                         src1=(fs.grad if fold else None), mode=ops.SRC_DIRECT, KH=2, KW=2, stride=1, pad=0, Ho=Ho, Wo=Wo,
                                         Cout=4 * C0, dtype=self.cdt, accumulate0=acc0)
                    qs.s2d_dgrad = 1
                    qs.weight = wb.data_ptr()                   # the ordinary data-gradient copies: the kernel addresses them per parity class
                    if fold:
                        qs.fold_weight = fs.meta["wb"].data_ptr()
                    if int(self.lib.stp_conv2d_halo_variant(C.byref(qs))) >= 0:
                        q, s2d = qs, True
                        if fold:
                            fs.meta["dgrad_folded"] = True
                            x.grad_writes += 1          # its share of x's gradient arrives with this launch
                if (not s2d and fs is not None and stride == 2 and k == 3 and not transpose and src1 is None and x_ng and fs.needs_grad and fs.grad_ready
                        and fs.meta.get("wb") is not None and fs.gradC == CoutB and (fs.H, fs.W) == (Ho, Wo)
                        and int(self.lib.stp_conv2d_fold_ok(C.byref(q)))):
                    # the shortcut's 1x1 / stride-2 data gradient = one more (centre) tap of this launch's (even, even) parity class
                    q.fold_src, q.fold_weight, q.fold_C = fs.grad.data_ptr(), fs.meta["wb"].data_ptr(), CoutB
                    fs.meta["dgrad_folded"] = True
                    x.grad_writes += 1          # its share of x's gradient arrives with this launch
                folded1 = False
                if (fs is not None and stride == 1 and k == 3 and pad == 1 and not transpose and not upsample and src1 is None and x_ng
                        and fs.needs_grad and fs.grad_ready and fs.meta.get("wb") is not None and (fs.H, fs.W) == (Ho, Wo)
                        and fs.meta.get("w_master", (0, 0, 0, 0))[2:] == (C0, 1) and self.dtype != "fp32" and os.environ.get("STP_HALO", "1") != "0"):
                    # the sibling 1x1 / stride-1 shortcut (first unit of ResNet18 / 34's stage 1): its dY is a second source of this launch
                    # whose centre tap carries the shortcut's weights (conv_halo.hip, FOLD1) - no separate launch accumulates into dX
                    q.fold_src, q.fold_weight, q.fold_C = fs.grad.data_ptr(), fs.meta["wb"].data_ptr(), fs.gradC
                    if int(self.lib.stp_conv2d_halo_variant(C.byref(q))) >= 0:
                        folded1 = True
                        fs.meta["dgrad_folded"] = True
                        x.grad_writes += 1          # its share of x's gradient arrives with this launch
                    else:
                        q.fold_src = q.fold_weight = None
                        q.fold_C = 0
                bnm = x.meta.get("bn")
                folded_up = False
                qC1 = C1
                if w4 is not None:
                    # conv3x3(concat(UpSampling2D(2)(x), skip)): the skip gradient is the plain 3x3 data gradient with the skip's rows
                    # of the flipped weight copy; the gradient of x is a 4x4 / stride-2 convolution of dY with the tap sums (w4) that
                    # lands on the low-resolution tensor directly - no high-resolution gradient, no stp_upsample2x_bwd
                    if s_ng:
                        q1 = ops.conv_params(dy, wb, d1, N=self.N, Hs0=Ho, Ws0=Wo, Hv=Ho, Wv=Wo, C0=CoutB, mode=ops.SRC_DIRECT, KH=k, KW=k,
                                             stride=1, pad=k - 1 - pad, Ho=Hv, Wo=Wv, Cout=C1, dtype=self.cdt, accumulate0=acc1)
                        q1.weight = wb.data_ptr() + C0 * k * k * CoutB * wb.element_size()
                        self._emit_conv(self.bwd, q1, {"layer": name, "pass": "dgrad", "flops": flops * C1 / float(C0 + C1),
                                                       "tile": int(self.lib.stp_conv2d_tile_for(C.byref(q1)))})
                    q = ops.conv_params(dy, w4, self._gradbuf(x), N=self.N, Hs0=Ho, Ws0=Wo, Hv=Ho, Wv=Wo, C0=CoutB, mode=ops.SRC_DIRECT,
                                        KH=4, KW=4, stride=2, pad=1, Ho=x.H, Wo=x.W, Cout=C0, dtype=self.cdt, accumulate0=int(x.grad_ready))
                    folded_up, qC1 = True, 0
                qflops = flops * (C0 - 0.0) / (C0 + C1) if w4 is not None else flops
                if upsample and x_ng and C1 == 0 and self.fold_upsample_grad:
                    # the small-channel kernel sums each 2x2 block in its epilogue: the hi-res gradient of the upsampled
                    # tensor is never written and stp_upsample2x_bwd disappears
                    q.dst_sum2x2 = 1
                    if self.lib.stp_conv2d_sc_eligible(C.byref(q)) and C0 % 4 == 0:
                        folded_up = True
                        q.dst0 = self._gradbuf(x).data_ptr()
                        q.accumulate0 = int(x.grad_ready)
                    else:
                        q.dst_sum2x2 = 0
                        # the halo kernel's summed epilogue (EP 3) for 64+ channels - in its fused form only (see the two-destination case below)
                        if (self.fuse_bn_backward and bnm is not None and x.meta.get("uses", 0) == 1 and not x.grad_ready
                                and self.slot_arena is None and os.environ.get("STP_HALO_FOLD_UP", "1") != "0"
                                and os.environ.get("STP_HALO", "1") != "0"):      # (the dispatcher honours STP_HALO=0: plan and dispatcher agree)
                            keep = (q.dst0, q.accumulate0)
                            q.dst_sum2x2, q.dst0, q.accumulate0 = 1, self._gradbuf(x).data_ptr(), 0
                            q.bnb_x, q.bnb_mean, q.bnb_rstd, q.bnb_gamma, q.bnb_beta, q.bnb_relu = bnm
                            if int(self.lib.stp_conv2d_halo_variant(C.byref(q))) >= 0:
                                folded_up = True
                            else:
                                q.dst_sum2x2, (q.dst0, q.accumulate0) = 0, keep
                                q.bnb_x = q.bnb_mean = q.bnb_rstd = q.bnb_gamma = q.bnb_beta = None
                                q.bnb_relu = 0
                two_dest = False
                if upsample and x_ng and C1 and w4 is None and self.fold_upsample_grad:
                    # ... and the wide-output kernel does the same for conv3x3(concat(UpSampling2D(2)(x), skip)): the first C0
                    # channels are summed into the low-resolution gradient, the skip's C1 channels stay at full resolution
                    q.dst_sum2x2 = 1
                    if self.lib.stp_conv2d_scw_eligible(C.byref(q)):
                        folded_up = two_dest = True
                        q.dst0 = self._gradbuf(x).data_ptr()
                        q.accumulate0 = int(x.grad_ready)
                    else:
                        q.dst_sum2x2 = 0
                        # ... and the halo kernel (64+ channel decoder stages, round 4): its epilogue sums the 2 x 2 blocks of the channel
                        # tiles of the upsampled source and runs the fused BatchNormalization backward on the LOW-resolution result
                        # (EP 3) - only in that fused form, i.e. when this launch completes the gradient of a BatchNormalization output
                        # that nothing else reads
                        if (self.fuse_bn_backward and bnm is not None and x.meta.get("uses", 0) == 1 and not x.grad_ready
                                and self.slot_arena is None and os.environ.get("STP_HALO_FOLD_UP", "1") != "0"
                                and os.environ.get("STP_HALO", "1") != "0"):      # (the dispatcher honours STP_HALO=0: plan and dispatcher agree)
                            keep = (q.dst0, q.accumulate0)
                            q.dst_sum2x2, q.dst0, q.accumulate0 = 1, self._gradbuf(x).data_ptr(), 0
                            q.bnb_x, q.bnb_mean, q.bnb_rstd, q.bnb_gamma, q.bnb_beta, q.bnb_relu = bnm
                            if int(self.lib.stp_conv2d_halo_variant(C.byref(q))) >= 0:
                                folded_up = two_dest = True
                            else:
                                q.dst_sum2x2, (q.dst0, q.accumulate0) = 0, keep
                                q.bnb_x = q.bnb_mean = q.bnb_rstd = q.bnb_gamma = q.bnb_beta = None
                                q.bnb_relu = 0
                if folded_up and d0_hires is not None:
                    # the full-resolution gradient of the upsampled tensor is never written: give its buffer back (33-134 MB per decoder
                    # stage at batch 16, 512 x 512 - it used to stay allocated for the life of the plan)
                    self._keep = [t for t in self._keep if t is not d0_hires]
                    d0_hires = None
                uses = x.meta.get("uses", 0)
                # the only consumer, or the LAST of several (every other consumer has already written or accumulated its
                # share, this data gradient accumulates on top): its epilogue sees the complete gradient of the BN output
                sole = uses == 1 and not q.accumulate0
                last = self.fuse_bn_backward_last and uses > 1 and x.grad_writes == uses - 1 and q.accumulate0 and not upsample
                if (folded1 or s2d) and uses == 2 and x.grad_writes == 1 and not q.accumulate0:
                    sole = True       # this launch and the sibling folded into it are the only two consumers: the gradient is complete
                if (self.fuse_bn_backward and bnm is not None and (sole or last) and (folded_up or not upsample) and (qC1 == 0 or two_dest)
                        and x_ng and C0 % 4 == 0):
                    q.bnb_x, q.bnb_mean, q.bnb_rstd, q.bnb_gamma, q.bnb_beta, q.bnb_relu = bnm
                    if self.slot_arena is not None and self.N * Hv * Wv <= self.bn_slots_max_rows and not two_dest:
                        # (the two-destination kernel has no slot form - stp_conv2d_scw_eligible was checked without them, and with
                        #  slots the launch would fall through to the generic kernel, which refuses dst_sum2x2: float partial sums there)
                        sp, sn = self._slots(C0)
                        q.stats_partial, q.stats_slots = sp, sn
                        x.meta["bnb_slots"] = (sp, sn)
                    else:
                        nfl = int(self.lib.stp_conv2d_stats_floats(C.byref(q)))
                        st = self._alloc((max(nfl, 4),), torch.float32)
                        q.stats_partial = st.data_ptr()
                        x.meta["bnb"] = (st, q)
                        gt, gcols = self._group_stats(q, st, C0)
                        if gt is not st:
                            x.meta["bnb"] = (gt, gcols)           # the pre-reduced table ([2][C0][columns / G]) and its column count
                self._emit_conv(self.bwd, q, {"layer": name, "pass": "dgrad", "flops": qflops,
                                              "tile": int(self.lib.stp_conv2d_tile_for(C.byref(q))), "s2d": s2d, "fold1": folded1})
                if upsample and x_ng and not folded_up:
                    acc_up = int(x.grad_ready)
                    done = (uses == 1 and not acc_up) or (self.fuse_bn_backward_last and uses > 1 and x.grad_writes == uses - 1 and acc_up)
                    ntl = int(self.lib.stp_upsample2x_bwd_bn_tiles(self.N, x.H, x.W, C0, C0, self.cdt)) if (
                        self.fuse_bn_backward and bnm is not None and done and self.slot_arena is None
                        and os.environ.get("STP_FUSE_UP_BN", "1") != "0") else 0
                    if ntl > 0:
                        # the upsampling gradient completes dY of a BatchNormalization output: mask + backward sums in the same pass
                        st = self._alloc((2 * C0 * ntl,), torch.float32)
                        self._emit(self.bwd, "stp_upsample2x_bwd_bn", d0.data_ptr(), self._gradbuf(x).data_ptr(), self.N, x.H, x.W,
                                   C0, C0, self.cdt, acc_up, bnm[0], bnm[1], bnm[2], bnm[3], bnm[4], bnm[5], st.data_ptr())
                        x.meta["bnb"] = (st, ntl)
                    else:
                        self._emit(self.bwd, "stp_upsample2x_bwd", d0.data_ptr(), self._gradbuf(x).data_ptr(), self.N, x.H, x.W,
                                   C0, C0, self.cdt, acc_up)
                if x_ng:
                    x.grad_ready = True
                if s_ng:
                    src1.grad_ready = True

        self._tape.append(back)
        return out

    def add(self, name, y, skip):
        """Keras ``Add()([y, skip])`` computed in place on y's buffer (y must have no other consumer): Linknet's
        decoder adds the encoder feature AFTER BatchNormalization + ReLU, so it cannot ride in a conv epilogue."""
        if (y.H, y.W, y.C) != (skip.H, skip.W, skip.C):
            raise StpShapeError("%s: cannot add %dx%dx%d and %dx%dx%d" % (name, y.H, y.W, y.C, skip.H, skip.W, skip.C))
        out = DT(name, self.N, y.H, y.W, y.C, y.buf, y.needs_grad or skip.needs_grad)
        out.gradC = y.gradC
        self.tensors[name] = out
        self._use(y, skip)
        if self.dry:
            return out
        self._emit(self.fwd, "stp_add_inplace", y.buf.data_ptr(), skip.buf.data_ptr(), y.rows * y.C, self.cdt)
        if not self.training:
            return out

        def back():
            if not out.needs_grad or not out.grad_ready:
                return
            dy = out.grad
            if y.needs_grad:
                y.grad, y.grad_ready = dy, True          # alias: read by y's backward, issued before any later writer below
            if skip.needs_grad:
                if not skip.grad_ready and skip.gradC == out.gradC:
                    skip.grad, skip.grad_ready = dy, True   # first gradient of the skip tensor: alias, later writers accumulate
                else:
                    self._emit(self.bwd, "stp_add_inplace", self._gradbuf(skip).data_ptr(), dy.data_ptr(), y.rows * out.gradC,
                               self.cdt)

        self._tape.append(back)
        return out

    def relu(self, name, x):
        """Keras ``Activation('relu')`` as a tensor of its own (DeepLab's xception blocks activate a tensor that its other
        consumers - the shortcut convolution, the Add - read raw, model.py:133-134): y = max(x, 0) through the BatchNormalization
        apply kernel with the identity affine; the gradient is masked in place by y > 0 and joins x's other gradients."""
        Cn = x.C
        out = self._new(name, x.H, x.W, Cn, x.needs_grad)
        out.gradC = x.gradC
        self._use(x)
        if self.dry:
            return out
        zero, one = self._alloc((Cn,), torch.float32), self._alloc((Cn,), torch.float32)
        zero.zero_(); one.fill_(1.0)
        self._emit(self.fwd, "stp_bn_apply", x.buf.data_ptr(), self.cdt, out.buf.data_ptr(), self.cdt, x.rows, Cn, Cn, zero.data_ptr(),
                   one.data_ptr(), None, None, 1, 0.0)
        if not self.training:
            return out

        def back():
            if not (x.needs_grad and out.grad_ready):
                return
            dy = out.grad
            if out.gradC != Cn:
                raise StpShapeError("%s: standalone ReLU needs an unpadded channel count" % name)
            self._before_inplace_write(dy)
            self._emit(self.bwd, "stp_relu_bwd", out.buf.data_ptr(), dy.data_ptr(), x.rows * Cn, self.cdt)
            if not x.grad_ready and x.gradC == out.gradC:
                x.grad, x.grad_ready = dy, True
            else:
                self._emit(self.bwd, "stp_add_inplace", self._gradbuf(x).data_ptr(), dy.data_ptr(), x.rows * Cn, self.cdt)

        self._tape.append(back)
        return out

    def upsample_add(self, name, x, m):
        """FPN top-down step ``Add()([x, UpSampling2D(2)(m)])`` in place on x's buffer (x must have no other consumer yet)."""
        if (x.H, x.W, x.C) != (2 * m.H, 2 * m.W, m.C):
            raise StpShapeError("%s: %dx%dx%d cannot take the 2x upsampling of %dx%dx%d" % (name, x.H, x.W, x.C, m.H, m.W, m.C))
        out = DT(name, self.N, x.H, x.W, x.C, x.buf, x.needs_grad or m.needs_grad)
        out.gradC = x.gradC
        self.tensors[name] = out
        self._use(x, m)
        if self.dry:
            return out
        self._emit(self.fwd, "stp_upsample2x_add", x.buf.data_ptr(), m.buf.data_ptr(), self.N, x.H, x.W, x.C, self.cdt)
        if not self.training:
            return out

        def back():
            if not out.needs_grad or not out.grad_ready:
                return
            dy = out.grad
            if m.needs_grad:
                self._emit(self.bwd, "stp_upsample2x_bwd", dy.data_ptr(), self._gradbuf(m).data_ptr(), self.N, m.H, m.W, m.C, out.gradC,
                           self.cdt, int(m.grad_ready))
                m.grad_ready = True
            if x.needs_grad:
                x.grad, x.grad_ready = dy, True

        self._tape.append(back)
        return out

    def upsample_sum(self, name, parts):
        """``sum_i ResizeImage(f_i, 'bilinear')(t_i)`` of up to four maps with equal channel counts (TF 1.x bilinear, integer factors; one fp32
        sum, one rounding): the resized pyramid terms of the PSPNet head, computed behind their 1x1 convolutions (nets.pspnet_resnet).  The
        gradient of each part is the bilinear-resize gradient of the sum's gradient."""
        if not 1 <= len(parts) <= 4:
            raise StpShapeError("%s: one to four parts" % name)
        Ho, Wo, Cn = parts[0][0].H * parts[0][1], parts[0][0].W * parts[0][1], parts[0][0].C
        if Ho != Wo or any((t.H * f, t.W * f, t.C) != (Ho, Wo, Cn) or t.H != t.W for t, f in parts):
            raise StpShapeError("%s: the parts must be square maps of equal channel count that resize to one size" % name)
        out = self._new(name, Ho, Wo, Cn, any(t.needs_grad for t, _ in parts))
        self._use(*[t for t, _ in parts])
        if self.dry:
            return out
        ptrs = [t.buf.data_ptr() for t, _ in parts] + [None] * (4 - len(parts))
        hs = [t.H for t, _ in parts] + [0] * (4 - len(parts))
        self._emit(self.fwd, "stp_upsample_sum", ptrs[0], ptrs[1], ptrs[2], ptrs[3], hs[0], hs[1], hs[2], hs[3], out.buf.data_ptr(), self.N, Ho, Wo, Cn,
                   self.cdt)
        if not self.training:
            return out

        def back():
            if not out.needs_grad or not out.grad_ready:
                return
            for t, f in parts:
                if not t.needs_grad:
                    continue
                wp, wb = self._scratch(self.lib.stp_resize_bilinear_bwd_workspace_bytes(self.N, t.H, t.W, t.C, f))
                self._emit(self.bwd, "stp_resize_bilinear_bwd", out.grad.data_ptr(), self._gradbuf(t).data_ptr(), self.N, t.H, t.W, t.C, f, out.gradC, 0,
                           self.cdt, int(t.grad_ready), wp, wb)
                t.grad_ready = True

        self._tape.append(back)
        return out

    def concat_resize(self, name, parts, nearest=False):
        """``Concatenate()([ResizeImage(f_i, interpolation)(t_i) ...])``: every part is resized (TF 1.x bilinear, or nearest;
        integer factor; 1 = copy) straight into its channel slice of the output."""
        Ho, Wo = parts[0][0].H * parts[0][1], parts[0][0].W * parts[0][1]
        if any((t.H * f, t.W * f) != (Ho, Wo) for t, f in parts):
            raise StpShapeError("%s: resized parts differ in size" % name)
        Ct = sum(t.C for t, _ in parts)
        out = self._new(name, Ho, Wo, Ct, any(t.needs_grad for t, _ in parts))
        self._use(*[t for t, _ in parts])
        if self.dry:
            return out
        off = 0
        for t, f in parts:
            self._emit(self.fwd, "stp_resize_nearest" if nearest else "stp_resize_bilinear", t.buf.data_ptr(), out.buf.data_ptr(), self.N, t.H,
                       t.W, t.C, f, Ct, off, self.cdt)
            off += t.C
        if not self.training:
            return out

        def back():
            if not out.needs_grad or not out.grad_ready:
                return
            o = 0
            for t, f in parts:
                if t.needs_grad and nearest:
                    self._emit(self.bwd, "stp_resize_nearest_bwd", out.grad.data_ptr(), self._gradbuf(t).data_ptr(), self.N, t.H, t.W,
                               t.C, f, out.gradC, o, self.cdt, int(t.grad_ready))
                    t.grad_ready = True
                elif t.needs_grad:
                    wp, wb = self._scratch(self.lib.stp_resize_bilinear_bwd_workspace_bytes(self.N, t.H, t.W, t.C, f))
                    self._emit(self.bwd, "stp_resize_bilinear_bwd", out.grad.data_ptr(), self._gradbuf(t).data_ptr(), self.N, t.H, t.W,
                               t.C, f, out.gradC, o, self.cdt, int(t.grad_ready), wp, wb)
                    t.grad_ready = True
                o += t.C

        self._tape.append(back)
        return out

    def resize(self, name, x, factor, nearest=False):
        """``ResizeImage(factor, 'bilinear' | 'nearest')`` of a tensor whose gradient carries padded channels (the class logits)."""
        out = self._new(name, x.H * factor, x.W * factor, x.C, x.needs_grad)
        out.gradC = x.gradC
        self._use(x)
        if self.dry:
            return out
        self._emit(self.fwd, "stp_resize_nearest" if nearest else "stp_resize_bilinear", x.buf.data_ptr(), out.buf.data_ptr(), self.N, x.H, x.W,
                   x.C, factor, x.C, 0, self.cdt)
        if not self.training:
            return out
        if not nearest:      # (softmax_loss may take this launch and its gradient over: stp_softmax_cce_dice_up)
            out.meta["resize_rec"] = (len(self.fwd) - 1, x, int(factor))

        def back():
            if not (x.needs_grad and out.grad_ready):
                return
            if nearest:
                self._emit(self.bwd, "stp_resize_nearest_bwd", out.grad.data_ptr(), self._gradbuf(x).data_ptr(), self.N, x.H, x.W, x.gradC,
                           factor, x.gradC, 0, self.cdt, int(x.grad_ready))
                x.grad_ready = True
                return
            wp, wb = self._scratch(self.lib.stp_resize_bilinear_bwd_workspace_bytes(self.N, x.H, x.W, x.gradC, factor))
            self._emit(self.bwd, "stp_resize_bilinear_bwd", out.grad.data_ptr(), self._gradbuf(x).data_ptr(), self.N, x.H, x.W, x.gradC,
                       factor, x.gradC, 0, self.cdt, int(x.grad_ready), wp, wb)
            x.grad_ready = True

        self._tape.append(back)
        return out

    def dwconv(self, name, x, k=3, stride=1, dilation=1, explicit_pad=False):
        """Keras ``DepthwiseConv2D(k, strides, padding='same', dilation_rate, use_bias=False)`` (DeepLab model.py:136, 255-259).
        The fp32 master kernel [k][k][C] is read directly by the kernels (kind "dw": Keras' (kh,kw,C,1) layout as stored)."""
        Cn = x.C
        if Cn % 4:
            raise StpShapeError("%s: depthwise convolution needs a multiple of 4 channels" % name)
        keff = (k - 1) * dilation + 1
        Ho, Wo = -(-x.H // stride), -(-x.W // stride)
        pt, pl = max((Ho - 1) * stride + keff - x.H, 0) // 2, max((Wo - 1) * stride + keff - x.W, 0) // 2
        if explicit_pad:
            # ZeroPadding2D((pad_beg, pad_end)) + 'valid' (DeepLab SepConv_BN with stride > 1, model.py:126-132): the padding does
            # not depend on the input size, unlike TF 'same'
            pt = pl = (keff - 1) // 2
            Ho, Wo = (x.H + (keff - 1) - keff) // stride + 1, (x.W + (keff - 1) - keff) // stride + 1
        w = self.param(name + "/depthwise_kernel", (k, k, Cn), "dw")
        out = self._new(name, Ho, Wo, Cn, x.needs_grad or w.trainable)
        self._use(x)
        self._dw_ws_bytes = max(self._dw_ws_bytes, int(self.lib.stp_dwconv_wgrad_workspace_bytes(Cn, k)))
        if self.dry:
            return out
        geo = (self.N, x.H, x.W, Cn, k, stride, pt, pl, dilation, Ho, Wo, self.cdt)
        self._emit(self.fwd, "stp_dwconv", x.buf.data_ptr(), self._pptr(w), out.buf.data_ptr(), *geo)
        if not self.training:
            return out

        def back():
            if not out.needs_grad or not out.grad_ready:
                return
            if w.trainable:
                self._emit(self.bwd, "stp_dwconv_wgrad", x.buf.data_ptr(), out.grad.data_ptr(), self._gptr(w), *geo, 0,
                           self.ws_dw.data_ptr(), self.ws_dw.numel() * 4)
            if x.needs_grad:
                self._emit(self.bwd, "stp_dwconv_dgrad", out.grad.data_ptr(), self._pptr(w), self._gradbuf(x).data_ptr(), *geo,
                           int(x.grad_ready))
                x.grad_ready = True

        self._tape.append(back)
        return out

    def dropout(self, name, x, rate, salt, spatial=False):
        """Keras ``Dropout(rate)`` / ``SpatialDropout2D(rate)`` (``spatial``: one decision per sample and channel): training =
        inverted dropout in place (mask = hash of the device step counter, which the first dropout of the plan ticks once per
        forward); inference = identity."""
        if not self.training:
            return x
        out = DT(name, self.N, x.H, x.W, x.C, x.buf, x.needs_grad)
        out.gradC = x.gradC
        self.tensors[name] = out
        self._use(x)
        if self.dry:
            return out
        if self.step_state is None:
            self.step_state = torch.zeros(2, dtype=torch.int32, device=self.device)
            self._keep.append(self.step_state)
            self._emit(self.fwd, "stp_counter_tick", self.step_state.data_ptr())
        cnt = x.rows * x.C
        if spatial:
            self._emit(self.fwd, "stp_dropout_spatial", x.buf.data_ptr(), x.buf.data_ptr(), self.N, x.H * x.W, x.C, float(rate),
                       self.step_state.data_ptr(), int(salt), self.cdt)
        else:
            self._emit(self.fwd, "stp_dropout", x.buf.data_ptr(), x.buf.data_ptr(), cnt, float(rate), self.step_state.data_ptr(), int(salt), self.cdt)

        def back():
            if not (x.needs_grad and out.grad_ready):
                return
            dy = out.grad
            if spatial:
                if out.gradC != x.C:
                    raise StpShapeError("%s: spatial dropout needs an unpadded channel count" % name)
                self._emit(self.bwd, "stp_dropout_spatial", dy.data_ptr(), dy.data_ptr(), self.N, x.H * x.W, x.C, float(rate),
                           self.step_state.data_ptr(), int(salt), self.cdt)
            else:
                self._emit(self.bwd, "stp_dropout", dy.data_ptr(), dy.data_ptr(), x.rows * out.gradC, float(rate), self.step_state.data_ptr(),
                           int(salt), self.cdt)
            x.grad, x.grad_ready = dy, True

        self._tape.append(back)
        return out

    def resize_ac(self, name, x, Ho, Wo):
        """``BilinearUpsampling`` (DeepLab model.py:56-100): tf.image.resize_bilinear(align_corners=True) to (Ho, Wo)."""
        out = self._new(name, Ho, Wo, x.C, x.needs_grad)
        out.gradC = x.gradC
        self._use(x)
        if self.dry:
            return out
        self._emit(self.fwd, "stp_resize_bilinear_ac", x.buf.data_ptr(), out.buf.data_ptr(), self.N, x.H, x.W, x.C, Ho, Wo, self.cdt)
        if not self.training:
            return out

        def back():
            if not (x.needs_grad and out.grad_ready):
                return
            self._emit(self.bwd, "stp_resize_bilinear_ac_bwd", out.grad.data_ptr(), self._gradbuf(x).data_ptr(), self.N, x.H, x.W, x.gradC,
                       Ho, Wo, self.cdt, int(x.grad_ready))
            x.grad_ready = True

        self._tape.append(back)
        return out

    def sigmoid_act(self, name, z):
        """The activation of DeepLab's last convolution as a layer (model.py:485): sigmoid for one class, channel softmax for
        2..32 classes."""
        out = self._new(name, z.H, z.W, z.C, z.needs_grad)
        out.gradC = z.gradC
        self._use(z)
        if self.dry:
            return out
        fn = "stp_sigmoid_act" if z.C == 1 else "stp_softmax_act"
        self._emit(self.fwd, fn, z.buf.data_ptr(), out.buf.data_ptr(), z.rows, z.C, z.C, z.C, self.cdt)
        if not self.training:
            return out

        def back():
            if not (z.needs_grad and out.grad_ready):
                return
            self._emit(self.bwd, fn + "_bwd", out.buf.data_ptr(), out.grad.data_ptr(), self._gradbuf(z).data_ptr(), z.rows, z.C,
                       z.C, z.gradC, self.cdt)
            z.grad_ready = True

        self._tape.append(back)
        return out

    def prob_loss(self, probs, target, w_ce, w_dice):
        """w*binary_crossentropy (one class) or w*categorical_crossentropy (class-index target) + w_dice*dice_loss on
        PROBABILITIES; seeds the backward pass."""
        if self.dry:
            return
        if self.training and self.loss_scale != 1.0:
            raise StpShapeError("the losses on probabilities (DeepLabV3) have no loss-scaling form: use loss_scale=1 (or bf16)")
        self.loss_scalars = self._alloc((12,), torch.float32)
        dp = self._gradbuf(probs) if self.training else None
        if probs.C == 1:
            self._emit(self.fwd, "stp_prob_bce_dice", probs.buf.data_ptr(), target.buf.data_ptr(), probs.rows, self.cdt, float(w_ce),
                       float(w_dice), self.loss_scalars.data_ptr(), dp.data_ptr() if dp is not None else None, probs.gradC,
                       self.ws_loss.data_ptr(), self.ws_loss.numel() * 4)
        else:
            self._emit(self.fwd, "stp_prob_cce_dice", probs.buf.data_ptr(), target.buf.data_ptr(), probs.rows, probs.C, probs.C, self.cdt,
                       float(w_ce), float(w_dice), self.loss_scalars.data_ptr(), dp.data_ptr() if dp is not None else None, probs.gradC,
                       self.ws_loss.data_ptr(), self.ws_loss.numel() * 4)
        probs.grad_ready = self.training

    def probs_out(self, probs):
        """Inference output when the model itself ends in probabilities: float32 copy."""
        if self.dry:
            return None
        if self.tdt == torch.float32:
            self.probs = probs.buf                      # already float32
            return self.probs
        self.probs = self._alloc((probs.N, probs.H, probs.W, probs.C), torch.float32)
        self._emit(self.fwd, "stp_cast_bf16_to_f32", probs.buf.data_ptr(), self.probs.data_ptr(), probs.rows * probs.C, 1.0)
        return self.probs

    def avgpool(self, name, x, k):
        """AveragePooling2D(pool_size = strides = k) with exact division (PSPNet pyramid levels)."""
        if x.H % k or x.W % k:
            raise StpShapeError("%s: %dx%d is not divisible by the pooling size %d" % (name, x.H, x.W, k))
        out = self._new(name, x.H // k, x.W // k, x.C, x.needs_grad)
        self._use(x)
        if self.dry:
            return out
        wp, wb = self._scratch(self.lib.stp_avgpool_workspace_bytes(self.N, x.H, x.W, x.C, k))
        self._emit(self.fwd, "stp_avgpool", x.buf.data_ptr(), out.buf.data_ptr(), self.N, x.H, x.W, x.C, k, self.cdt, wp, wb)
        if not self.training:
            return out

        def back():
            if not (x.needs_grad and out.grad_ready):
                return
            self._emit(self.bwd, "stp_avgpool_bwd", out.grad.data_ptr(), self._gradbuf(x).data_ptr(), self.N, x.H, x.W, x.C, k, self.cdt,
                       int(x.grad_ready))
            x.grad_ready = True

        self._tape.append(back)
        return out

    def avgpool_pyramid(self, names, x, ks):
        """The AveragePooling2D(k) of PSPNet's pyramid levels (one per name / window size) of ONE tensor: where the windows nest (every k a
        multiple of the smallest one) a single pass over ``x`` (stp_avgpool_pyramid) and a single pass over its gradient instead of one per
        level; otherwise the separate launches of ``avgpool``.  Returns the pooled tensors in the order of ``names``."""
        order = sorted(range(len(ks)), key=lambda i: ks[i])
        kk = [int(ks[i]) for i in order] + [0] * (4 - len(ks))
        if (len(ks) < 2 or len(ks) > 4 or any(x.H % k or x.W % k for k in ks)
                or not self.lib.stp_avgpool_pyramid_ok(self.N, x.H, x.W, x.C, kk[0], kk[1], kk[2], kk[3], self.cdt)):
            return [self.avgpool(nm, x, k) for nm, k in zip(names, ks)]
        outs = [self._new(nm, x.H // k, x.W // k, x.C, x.needs_grad) for nm, k in zip(names, ks)]
        for _ in outs:
            self._use(x)
        if self.dry:
            return outs
        so = [outs[i] for i in order]
        wp, wb = self._scratch(self.lib.stp_avgpool_pyramid_workspace_bytes(self.N, x.H, x.W, x.C, kk[0], self.cdt))
        yp = [t.buf.data_ptr() for t in so] + [None] * (4 - len(so))
        self._emit(self.fwd, "stp_avgpool_pyramid", x.buf.data_ptr(), yp[0], yp[1], yp[2], yp[3], kk[0], kk[1], kk[2], kk[3], self.N, x.H, x.W,
                   x.C, self.cdt, wp, wb)
        if not self.training:
            return outs

        def back():
            if not x.needs_grad:
                return
            ready = [t for t in so if t.grad_ready]
            if len(ready) == len(so):
                gp = [t.grad.data_ptr() for t in so] + [None] * (4 - len(so))
                self._emit(self.bwd, "stp_avgpool_pyramid_bwd", gp[0], gp[1], gp[2], gp[3], kk[0], kk[1], kk[2], kk[3], self._gradbuf(x).data_ptr(),
                           self.N, x.H, x.W, x.C, self.cdt, int(x.grad_ready))
                x.grad_ready = True
                return
            for t, k in zip(so, kk):      # (a level without a gradient: the separate launches for the others)
                if t.grad_ready:
                    self._emit(self.bwd, "stp_avgpool_bwd", t.grad.data_ptr(), self._gradbuf(x).data_ptr(), self.N, x.H, x.W, x.C, k, self.cdt,
                               int(x.grad_ready))
                    x.grad_ready = True

        self._tape.append(back)
        return outs

    def maxpool_k(self, name, x, k):
        """MaxPooling2D(pool_size = strides = k) (PSPNet ``psp_pooling_type: max``)."""
        if x.H % k or x.W % k:
            raise StpShapeError("%s: %dx%d is not divisible by the pooling size %d" % (name, x.H, x.W, k))
        out = self._new(name, x.H // k, x.W // k, x.C, x.needs_grad)
        self._use(x)
        if self.dry:
            return out
        idx = self._alloc((self.N, x.H // k, x.W // k, x.C), torch.int32) if self.training else None
        self._emit(self.fwd, "stp_maxpool_k", x.buf.data_ptr(), out.buf.data_ptr(), idx.data_ptr() if idx is not None else None, self.N, x.H,
                   x.W, x.C, k, self.cdt)
        if not self.training:
            return out

        def back():
            if not (x.needs_grad and out.grad_ready):
                return
            self._emit(self.bwd, "stp_maxpool_k_bwd", idx.data_ptr(), out.grad.data_ptr(), self._gradbuf(x).data_ptr(), self.N, x.H, x.W, x.C, k,
                       self.cdt, int(x.grad_ready))
            x.grad_ready = True

        self._tape.append(back)
        return out

    def maxpool2(self, name, x):
        """MaxPooling2D(2, 2) without padding (VGG blocks)."""
        if x.H % 2 or x.W % 2:
            raise StpShapeError("%s: 2x2 pooling needs even height/width" % name)
        out = self._new(name, x.H // 2, x.W // 2, x.C, x.needs_grad)
        self._use(x)
        if self.dry:
            return out
        idx = self._alloc((self.N, x.H // 2, x.W // 2, x.C), torch.uint8) if self.training else None
        self._emit(self.fwd, "stp_maxpool2x2", x.buf.data_ptr(), out.buf.data_ptr(), idx.data_ptr() if idx is not None else None,
                   self.N, x.H, x.W, x.C, self.cdt)

        def back():
            if not (x.needs_grad and out.grad_ready):
                return
            self._emit(self.bwd, "stp_maxpool2x2_bwd", idx.data_ptr(), out.grad.data_ptr(), self._gradbuf(x).data_ptr(), self.N,
                       x.H, x.W, x.C, self.cdt, int(x.grad_ready))
            x.grad_ready = True

        if self.training:
            self._tape.append(back)
        return out

    def maxpool(self, name, x):
        Ho, Wo = (x.H + 2 - 3) // 2 + 1, (x.W + 2 - 3) // 2 + 1
        out = self._new(name, Ho, Wo, x.C, x.needs_grad)
        self._use(x)
        if self.dry:
            return out
        idx = self._alloc((self.N, Ho, Wo, x.C), torch.uint8) if self.training else None
        last = self.fwd[-1] if self.fwd else None
        if (self.training and self.dtype != "fp32" and last is not None and last[2] == "stp_bn_apply" and last[1][2] == x.buf.data_ptr()
                and last[1][1] == last[1][3] == self.cdt and last[1][5] == last[1][6] == x.C and last[1][12] == 0.0 and x.C % 8 == 0
                and x.H % 2 == 0 and x.W % 2 == 0 and x.meta.get("apply_rec") in (None, last) and not x.meta.get("sc_consumers")
                and not x.meta.get("halo_consumers") and os.environ.get("STP_FUSE_BN_POOL_FWD", "1") != "0"):
            # the BatchNormalization apply that produced x is the previous launch (the stem: bn0 -> relu0 -> pooling0): ONE launch normalises,
            # stores x (the decoder's skip tensor) and pools - the 134 MB pre-normalisation tensor is read once instead of twice
            xa = last[1]
            self.fwd.pop()
            x.meta.pop("apply_rec", None)      # (the launch no longer exists on its own: nothing may drop or move it)
            self._emit(self.fwd, "stp_bn_apply_maxpool3x3s2", xa[0], x.buf.data_ptr(), out.buf.data_ptr(), idx.data_ptr() if idx is not None else None,
                       self.N, x.H, x.W, x.C, self.cdt, xa[7], xa[8], xa[9], xa[10], xa[11])
        else:
            self._emit(self.fwd, "stp_maxpool3x3s2", x.buf.data_ptr(), out.buf.data_ptr(), idx.data_ptr() if idx is not None else None,
                       self.N, x.H, x.W, x.C, self.cdt)

        def back():
            if not (x.needs_grad and out.grad_ready):
                return
            acc = int(x.grad_ready)
            bnm, uses = x.meta.get("bn"), x.meta.get("uses", 0)
            done = (uses == 1 and not acc) or (self.fuse_bn_backward_last and uses > 1 and x.grad_writes == uses - 1 and acc)
            ntl = int(self.lib.stp_maxpool3x3s2_bwd_bn_tiles(self.N, x.H, x.W, x.C, self.cdt)) if (
                self.fuse_bn_backward and bnm is not None and done and self.slot_arena is None
                and os.environ.get("STP_FUSE_POOL_BN", "0") == "1") else 0   # measured: 10.06 -> 10.14 ms when on (the gather + x read +
            #                                                                    reduce in one kernel runs at half the rate of the pair)
            if ntl > 0:
                # the pool gradient completes dY of a BatchNormalization output (bn0: the other consumer is a decoder skip)
                st = self._alloc((2 * x.C * ntl,), torch.float32)
                self._emit(self.bwd, "stp_maxpool3x3s2_bwd_bn", idx.data_ptr(), out.grad.data_ptr(), self._gradbuf(x).data_ptr(), self.N,
                           x.H, x.W, x.C, self.cdt, acc, bnm[0], bnm[1], bnm[2], bnm[3], bnm[4], bnm[5], st.data_ptr())
                x.meta["bnb"] = (st, ntl)
            else:
                self._emit(self.bwd, "stp_maxpool3x3s2_bwd", idx.data_ptr(), out.grad.data_ptr(), self._gradbuf(x).data_ptr(), self.N,
                           x.H, x.W, x.C, self.cdt, acc)
            x.grad_ready = True

        if self.training:
            self._tape.append(back)
        return out

    def sigmoid_loss(self, logits, target, w_bce, w_dice, w_iou=0.0, w_jaccard=0.0, w_focal=0.0, w_lovasz=0.0):
        """sigmoid + w_bce*binary_crossentropy + w_dice*dice_loss [+ w*iou_loss + w*jaccard_loss + w*focal_loss + w*lovasz_loss,
        the rest of the registry at reference segmentation.py:15-22]; seeds the backward pass."""
        if logits.C != 1:
            raise StpShapeError("binary loss expects one class")
        if self.dry:
            return
        self.loss_scalars = self._alloc((16,), torch.float32)
        self.loss_scalars.zero_()
        count = logits.rows
        dl = self._gradbuf(logits) if self.training else None
        if w_iou or w_jaccard or w_focal:
            import ctypes
            self._loss_weights = (ctypes.c_float * 5)(w_bce, w_dice, w_iou, w_jaccard, w_focal)     # host array read at launch
            self._emit(self.fwd, "stp_sigmoid_loss_ex", logits.buf.data_ptr(), target.buf.data_ptr(), count, self.cdt,
                       ctypes.addressof(self._loss_weights), self.loss_scalars.data_ptr(), dl.data_ptr() if dl is not None else None,
                       logits.gradC, float(self.loss_scale), self.ws_loss.data_ptr(), self.ws_loss.numel() * 4)
        else:
            self._emit(self.fwd, "stp_sigmoid_bce_dice", logits.buf.data_ptr(), target.buf.data_ptr(), count, self.cdt, float(w_bce),
                       float(w_dice), self.loss_scalars.data_ptr(), dl.data_ptr() if dl is not None else None, logits.gradC,
                       float(self.loss_scale), self.ws_loss.data_ptr(), self.ws_loss.numel() * 4)
        if self.training and self.dls is not None:
            self._emit(self.fwd, "stp_scale_by_device", dl.data_ptr(), count * logits.gradC, self.cdt, self.dls.data_ptr(), self.dls.data_ptr() + 16)
        # the class convolution reads its bias gradient from the gradient kernel's per-workgroup sums (not when another launch
        # adds to / rescales the gradient afterwards)
        logits.meta["loss_bias_grad"] = bool(self.training and not w_lovasz and self.dls is None)
        if w_lovasz:      # per-image Lovasz hinge: ADDS to scalars[0] and to the gradient the launch above wrote
            if self.training and self.loss_scale != 1.0:
                raise StpShapeError("lovasz_loss has no loss-scaling form: use loss_scale=1 (or bf16) with it")
            nbytes = int(self.lib.stp_lovasz_workspace_bytes(count, self.N))
            if nbytes <= 0:
                raise StpShapeError("lovasz_loss: the sort workspace cannot be sized (no device)")
            self.ws_lovasz = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._emit(self.fwd, "stp_lovasz_hinge", logits.buf.data_ptr(), target.buf.data_ptr(), self.N, count // self.N, self.cdt,
                       float(w_lovasz), self.loss_scalars.data_ptr(), dl.data_ptr() if dl is not None else None, logits.gradC,
                       self.ws_lovasz.data_ptr(), nbytes)
        logits.grad_ready = self.training

    def softmax_loss(self, logits, target, w_cce, w_dice):
        """channel softmax + w_cce*categorical_crossentropy + w_dice*dice_loss (target = class index per pixel)."""
        if logits.C < 2:
            raise StpShapeError("categorical loss expects at least two classes")
        if self.dry:
            return
        self.loss_scalars = self._alloc((12,), torch.float32)
        rec = logits.meta.get("resize_rec")
        if (self.training and rec is not None and self.fwd[rec[0]][2] == "stp_resize_bilinear" and rec[1].needs_grad
                and not rec[1].grad_ready and target.buf.data_ptr() % 4 == 0
                and self.lib.stp_softmax_cce_dice_up_ok(rec[2], logits.C, self.cdt)):
            # the logits are a bilinear resize of the class convolution's output (PSPNet, FPN) and only this loss reads them: both loss passes
            # interpolate from the low-resolution tensor, the gradient pass reduces straight into its gradient - the resized logits, their
            # gradient, the resize launch, stp_scale_by_device and stp_resize_bilinear_bwd leave the step (the launch record is kept for
            # HipSegModel.logits(), which materialises the resized tensor on demand)
            idx, lo, f = rec
            logits.meta["deferred"] = (self.fwd[idx][0], self.fwd[idx][1], self.fwd[idx][2])
            self.fwd[idx] = (None, (), "fused:resize->loss", None)
            nb = int(self.lib.stp_softmax_cce_dice_up_corner_bytes(self.N, lo.H, lo.W, logits.C))
            corners = self._alloc((nb // 4,), torch.float32)
            dlow = self._gradbuf(lo)
            dls = self.dls.data_ptr() if self.dls is not None else None
            self._emit(self.fwd, "stp_softmax_cce_dice_up", lo.buf.data_ptr(), target.buf.data_ptr(), self.N, lo.H, lo.W, f, logits.C, lo.C,
                       self.cdt, float(w_cce), float(w_dice), self.loss_scalars.data_ptr(), dlow.data_ptr(), lo.gradC, float(self.loss_scale),
                       dls, (dls + 16) if dls is not None else None, self.ws_loss.data_ptr(), self.ws_loss.numel() * 4, corners.data_ptr(), nb)
            lo.grad_ready = True
            logits.grad_ready = False
            logits.meta["fused_into_loss"] = True
            return
        dl = self._gradbuf(logits) if self.training else None
        self._emit(self.fwd, "stp_softmax_cce_dice", logits.buf.data_ptr(), target.buf.data_ptr(), logits.rows, logits.C, logits.C,
                   self.cdt, float(w_cce), float(w_dice), self.loss_scalars.data_ptr(), dl.data_ptr() if dl is not None else None,
                   logits.gradC, float(self.loss_scale), self.ws_loss.data_ptr(), self.ws_loss.numel() * 4)
        if self.training and self.dls is not None:
            self._emit(self.fwd, "stp_scale_by_device", dl.data_ptr(), logits.rows * logits.gradC, self.cdt, self.dls.data_ptr(), self.dls.data_ptr() + 16)
        logits.grad_ready = self.training

    def softmax_out(self, logits):
        """Activation('softmax') of the head for inference: float32 probabilities."""
        if self.dry:
            return None
        probs = self._alloc((logits.N, logits.H, logits.W, logits.C), torch.float32)
        self._emit(self.fwd, "stp_softmax", logits.buf.data_ptr(), probs.data_ptr(), logits.rows, logits.C, logits.C, self.cdt)
        self.probs = probs
        return probs

    def sigmoid_out(self, logits):
        """Activation('sigmoid') of the head for inference: float32 probabilities."""
        if self.dry:
            return None
        probs = self._alloc((logits.N, logits.H, logits.W, logits.C), torch.float32)
        self._emit(self.fwd, "stp_sigmoid", logits.buf.data_ptr(), probs.data_ptr(), logits.rows * logits.C, self.cdt)
        self.probs = probs
        return probs

    # ------------------------------------------------------------------ execution
    def run(self, launches):
        if self.device.type != "cuda":
            raise _lib.StpError("plans execute on the GPU only (no CPU fallback)")
        main = torch.cuda.current_stream()
        st = main.cuda_stream
        side, forked = None, False
        for fn, args, name, meta in launches:
            if fn is None:                      # stream markers of the weight-gradient side chain
                if name == "fork" and self.side_stream_wgrad:
                    side = self._side_stream()
                    side.wait_stream(main)      # everything issued so far (dY of this layer) is visible to the side chain
                    forked = True
                elif name == "join" and forked:
                    main.wait_stream(side)
                    forked = False
                continue
            s = st
            if meta is not None and meta.get("stream") and self.side_stream_wgrad:
                if not forked:                  # a launch list cut between fork and its side launches (graph segments)
                    side = self._side_stream()
                    side.wait_stream(main)
                    forked = True
                s = side.cuda_stream
            rc = fn(*args, s)
            if rc != 0:
                _lib.check(rc, name)
        if forked:
            main.wait_stream(side)

    # launches that read neither the weight compute copies nor the slot arena: they may run next to the weight preparation
    _PREP_FREE = ("stp_bn_stats", "stp_bn_apply", "stp_bn_inference", "stp_counter_tick")

    def run_prep_fwd(self):
        """``run(prep); run(fwd)`` with the weight compute copies (they depend on the master parameters only) on the side stream, next to
        the input BatchNormalization of the raw image (statistics + normalisation: ~25 us of small launches against ~55 us of weight
        copies on U-Net/ResNet34) - joined before the first launch that reads a weight.  Captured into the step's hipGraph as a
        fork / join.  OPT-IN (STP_PREP_SIDE=1): measured SLOWER - 7.27 vs 7.13 ms per step on one box (profiles/r04j_schedule_ab.txt): a
        fork / join inside a hipGraph costs far more (~0.14 ms here) than the 55 us of copies it hides."""
        k = 0
        while k < len(self.fwd) and self.fwd[k][2] in self._PREP_FREE:
            k += 1
        if os.environ.get("STP_PREP_SIDE", "0") != "1" or not self.prep or k == 0 or self.device.type != "cuda":
            self.run(self.prep)
            self.run(self.fwd)
            return
        main, side = torch.cuda.current_stream(), self._side_stream()
        side.wait_stream(main)
        for fn, args, name, _meta in self.prep:
            rc = fn(*args, side.cuda_stream)
            if rc != 0:
                _lib.check(rc, name)
        self.run(self.fwd[:k])
        main.wait_stream(side)
        self.run(self.fwd[k:])

    # loss launches whose third argument is the element / pixel count of the batch ([N, ...] -> the first n_valid samples)
    LOSS_LAUNCHES = ("stp_sigmoid_bce_dice", "stp_softmax_cce_dice", "stp_prob_bce_dice", "stp_sigmoid_loss_ex", "stp_prob_cce_dice")

    def rerun_loss(self, n_valid):
        """Re-evaluates the loss / metric reduction over the first ``n_valid`` samples only (an evaluation batch whose tail
        was filled by wrapping around: the duplicates must not enter val_loss / dice - Keras evaluates a short last batch
        as it is).  Samples are the slowest dimension of every tensor, so the real ones are a prefix of the element range.
        ``stp_lovasz_hinge`` ADDS its term to the scalars the first launch wrote, so it is re-run after it with
        ``images = n_valid`` (its third argument)."""
        n_valid = int(n_valid)
        if not 0 < n_valid <= self.N:
            raise ValueError("n_valid out of range")
        st = torch.cuda.current_stream().cuda_stream
        done = False
        for fn, args, name, _meta in self.fwd:
            if name in self.LOSS_LAUNCHES and not done:
                a = list(args)
                a[2] = args[2] // self.N * n_valid        # element / pixel count: [N, ...] -> [n_valid, ...]
                _lib.check(fn(*a, st), name)
                done = True
            elif name == "stp_lovasz_hinge" and done:
                a = list(args)
                a[2] = n_valid                            # images
                _lib.check(fn(*a, st), name)
        if not done:
            raise _lib.StpError("the plan has no loss launch")

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def init_states(self):
        for name, (off, numel, init) in self.states.items():
            self.S[off:off + numel] = init

    def set_trainable_mask(self):
        self.mask.fill_(1)
        for info in self.params.values():
            if not info.trainable:
                self.mask[info.offset:info.offset + info.numel] = 0
