"""Model object of the HIP backend: what ``PipelineConfig.createNet()`` hands back in place of a
``keras.Model`` (plugin contract, SURVEY 8b: ``predict``, ``load_weights``, ``save_weights`` as used
at reference ``segmentation_pipeline/segmentation.py:44,143,152,236``, plus ``train_on_batch``).

Weights cross this boundary in Keras layouts (conv kernels HWIO, BN vectors [C]) under the
layer names of segmentation_models / classification_models; on the device they live in one
flat fp32 arena (kernels as OHWI) so the optimizer and the gradient all-reduce are single
streaming launches.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import distributed, graph, nets

# hipGraph capture must not be invalidated by HIP calls of OTHER threads (the RCCL watchdog of torch.distributed polls
# events while a rank captures its step): thread-local capture mode
CAPTURE_MODE = "thread_local"

SCALAR_NAMES = ("loss", "binary_crossentropy", "dice_loss", "dice", "binary_accuracy")


EXTENDED_LOSSES = ("iou_loss", "jaccard_loss", "focal_loss", "lovasz_loss")      # sigmoid head only (stp_sigmoid_loss_ex, stp_lovasz_hinge)


def parse_loss(spec, classes=1, architecture=None):
    """``"binary_crossentropy+0.1*dice_loss"`` -> (w_ce, w_dice) or, when the spec names one of the other registry entries
    of reference segmentation.py:15-22, (w_ce, w_dice, w_iou, w_jaccard, w_focal, w_lovasz)  (grammar: reference README.md:210-214).
    The cross-entropy term is ``binary_crossentropy`` for the 1-class sigmoid head and ``categorical_crossentropy`` for
    the softmax head (schemas/segmentation.raml:12-21)."""
    ce = "binary_crossentropy" if classes == 1 else "categorical_crossentropy"
    w = {ce: 0.0, "dice_loss": 0.0}
    if classes == 1 and architecture != "DeepLabV3":
        w.update((k, 0.0) for k in EXTENDED_LOSSES)
    for term in str(spec).split("+"):
        term = term.strip()
        if "*" in term:
            k, name = term.split("*", 1)
            k, name = float(k), name.strip()
        else:
            k, name = 1.0, term
        if name not in w:
            raise ValueError("loss %r is not available in the HIP backend (have: %s)" % (name, ", ".join(sorted(w))))
        w[name] += k
    if any(w.get(k) for k in EXTENDED_LOSSES):
        return (w[ce], w["dice_loss"]) + tuple(w[k] for k in EXTENDED_LOSSES)
    return w[ce], w["dice_loss"]


class HipSegModel(object):
    def __init__(self, architecture="Unet", backbone="resnet34", input_shape=(512, 512, 3), classes=1, activation="sigmoid",
                 batch=16, dtype="bf16", loss="binary_crossentropy", optimizer="Adam", lr=1e-3, freeze_encoder=False,
                 decoder_filters=(256, 128, 64, 32, 16), clipnorm=None, clipvalue=None, use_graph=True, device="cuda",
                 opt_kwargs=None, seed=42, decoder_block_type="upsampling", net_kwargs=None, loss_scale=None):
        if architecture not in nets.NETWORKS:
            raise ValueError("Unknown architecture")
        if backbone not in nets.known_backbones() or (backbone in nets.VGG_BLOCKS and architecture not in ("Unet", "Linknet", "FPN", "PSPNet")) \
                or ((backbone in ("mobilenetv2", "xception")) != (architecture == "DeepLabV3")):   # VGG: every segmentation_models architecture; MobileNetV2 / Xception: DeepLabV3 only
            raise ValueError("Unknown backbone")
        if not ((classes == 1 and activation in ("sigmoid", None)) or (2 <= classes <= 32 and activation == "softmax")):
            raise ValueError("the HIP backend trains 1-class sigmoid heads and 2..32-class softmax heads")
        self.architecture, self.backbone = architecture, backbone
        self.H, self.W, self.in_ch = int(input_shape[0]), int(input_shape[1]), int(input_shape[2])
        self.classes, self.batch, self.dtype = classes, int(batch), dtype
        self.decoder_filters = tuple(decoder_filters)
        if decoder_block_type not in ("upsampling", "transpose") or (decoder_block_type == "transpose" and architecture not in ("Unet", "Linknet")):
            raise ValueError("decoder_block_type %r is not available for %s" % (decoder_block_type, architecture))
        self.decoder_block_type = decoder_block_type
        # architecture-specific graph options (FPN: pyramid_block_filters, segmentation_block_filters; PSPNet: downsample_factor,
        # psp_conv_filters - schemas/segmentation.raml:179-249), passed to the network definition as keywords
        self.net_kwargs = dict(net_kwargs or {})
        self.loss_w = parse_loss(loss, classes, architecture)
        self.optimizer = optimizer.lower()
        if self.optimizer not in ("adam", "sgd", "rmsprop", "nadam"):
            raise ValueError("optimizer %r is not available in the HIP backend (have: SGD, Adam, RMSprop, Nadam)" % optimizer)
        self.opt_kwargs = dict(opt_kwargs or {})
        self.clipnorm = float(clipnorm) if clipnorm else 0.0
        self.clipvalue = float(clipvalue) if clipvalue else 0.0
        self.use_graph = use_graph
        self.device = torch.device(device)
        self.reducer = None
        self.dp_overlap = False
        self._segments = None
        self._works = []
        self._graphs = None
        self.plan = graph.Plan(self.batch, dtype, device, training=True)
        # fp16 storage (BASELINE configs[3]): static loss scaling - the loss kernels seed the backward with loss_scale * dL/dlogits
        # (their grad_scale argument), every gradient in the arena is linear in it, and the optimizers' device scalar gscale
        # carries 1/loss_scale (times the data-parallel mean and the clipnorm factor).  2^14 keeps the 1/(N*H*W) BCE gradient
        # of a 16x512x512 batch (2.4e-7, below fp16's normal range) at 4e-3.  The losses without a grad_scale argument (on
        # probabilities: DeepLabV3; lovasz) run unscaled.
        scalable = architecture != "DeepLabV3" and not (len(self.loss_w) == 6 and self.loss_w[5])      # (.., w_lovasz)
        if loss_scale is None:
            loss_scale = 16384.0 if (dtype == "fp16" and scalable) else 1.0
            if dtype == "fp16" and not scalable:
                import warnings
                warnings.warn("fp16 storage without loss scaling: the losses on probabilities (DeepLabV3) and lovasz_loss have no scaled form, "
                              "their 1/(N*H*W) gradients sit in fp16's subnormal range - use dtype bf16 for this model / loss", RuntimeWarning)
        self.loss_scale = float(loss_scale)
        if self.loss_scale <= 0:
            raise ValueError("loss_scale must be positive")
        self.plan.loss_scale = self.loss_scale
        # DYNAMIC re-scaling on top of the static scale (fp16 only; STP_DYNAMIC_LOSS_SCALE=0 keeps the static scale + skip-on-overflow of
        # round 3): a device multiplier m (float[8] record, include/stp_hip.h) applied to the loss gradient right after the loss kernel,
        # halved when a step is skipped, doubled after `interval` clean steps - all inside the captured step.
        self.dls = None
        if dtype == "fp16" and self.loss_scale != 1.0 and os.environ.get("STP_DYNAMIC_LOSS_SCALE", "1") != "0":
            interval = float(os.environ.get("STP_LOSS_SCALE_INTERVAL", "2000"))
            # largest multiplier: 1 - the schedule only backs off after a non-finite gradient and recovers to the static scale.  The fp16
            # build saturates every 16-bit store at +-65504, so an overflowing activation gradient is clamped, never inf: growth beyond
            # the static scale could not be policed by the non-finite guard and would drift into silently clipped gradients (advisor,
            # round 4).  STP_LOSS_SCALE_MAX_MULT=<power of two> restores a growing schedule for experiments.
            mmax = float(os.environ.get("STP_LOSS_SCALE_MAX_MULT", "1"))
            self.dls = torch.tensor([1.0, 0.0, interval, 2.0 ** -14, 1.0, mmax, 0.0, 0.0], dtype=torch.float32, device=self.device)
            self.plan.dls = self.dls
        if freeze_encoder:
            self.plan.frozen_prefixes = nets.ENCODER_PREFIXES
        self.plan.define(self._net(True))
        p = self.plan
        p.init_states()
        p.set_trainable_mask()
        n = p.P.numel()
        self.lr = torch.tensor([float(lr)], dtype=torch.float32, device=self.device)
        self.opt_state = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.gscale = torch.ones(2, dtype=torch.float32, device=self.device)       # [gradient scale (<= 0: skip the step), skipped steps]
        self.gscale[1] = 0.0
        self.ws_norm = torch.empty(1024, dtype=torch.float32, device=self.device)
        self.m = self.v = self.vel = self.opt_fstate = None
        if self.optimizer in ("adam", "nadam"):
            self.m = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.v = torch.zeros(n, dtype=torch.float32, device=self.device)
            if self.optimizer == "nadam":
                self.opt_fstate = torch.zeros(8, dtype=torch.float32, device=self.device)
                self.opt_fstate[0] = 1.0      # m_schedule
        elif self.optimizer == "rmsprop":
            self.m = torch.zeros(n, dtype=torch.float32, device=self.device)   # the squared-gradient accumulator
        elif self.opt_kwargs.get("momentum", 0.0):
            self.vel = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.dp_scale = 1.0
        self.gscale[0] = 1.0 / self.loss_scale
        self._build_opt(use_gscale=self.loss_scale != 1.0)
        self._infer = None
        self.init_weights(seed)

    # ------------------------------------------------------------------ construction helpers
    def _net(self, training, with_loss=None):
        with_loss = training if with_loss is None else with_loss

        def fn(plan):
            kw = {"decoder_block_type": self.decoder_block_type} if self.architecture in ("Unet", "Linknet") else dict(self.net_kwargs)
            if self.architecture == "DeepLabV3" and self.backbone != "xception":
                kw.pop("OS", None)                                   # (the output stride applies to the xception backbone only, model.py:296-297)
            logits = nets.NETWORKS[self.architecture](plan, self.backbone, self.H, self.W, self.in_ch, self.classes,
                                                      self.decoder_filters, self.loss_w, with_loss=with_loss, **kw)
            if not with_loss and self.architecture != "DeepLabV3":      # (DeepLab's graph ends in probabilities itself)
                (plan.sigmoid_out if self.classes == 1 else plan.softmax_out)(logits)
            return logits
        return fn

    def eval_plan(self):
        """Inference-phase plan WITH the loss/metric reduction (validation pass of fit()); shares weights."""
        if getattr(self, "_eval", None) is None:
            ep = graph.Plan(self.batch, self.dtype, str(self.device), training=False)
            ep.define(self._net(False, with_loss=True), share=self.plan)
            self._eval = ep
        return self._eval

    def set_data_parallel(self, reducer, overlap=True):
        """Attaches a gradient reducer (distributed.GradReducer): gradients are SUM-all-reduced between
        the backward and optimizer graphs and the 1/world mean is folded into the optimizer.

        ``overlap``: the backward is cut into segments after which a range of the gradient arena is final (the arena
        is written from its tail towards its head, Plan.bwd_marks); the all-reduce of that range is issued
        asynchronously as soon as its segment has been launched, so RCCL runs under the remaining backward GEMMs.
        ``overlap=True`` uses distributed.two_phase_bounds; ``overlap="buckets"`` keeps the reducer's own buckets."""
        self.reducer = reducer
        self.dp_overlap = bool(overlap) and self.plan.bwd_monotone
        if self.dp_overlap and overlap != "buckets":
            reducer.set_bounds(distributed.two_phase_bounds(self.plan.bwd_marks, self.plan.G.numel()))
        self._segments = None
        self._works = []
        self.dp_scale = float(reducer.scale)
        self.gscale[0] = self.dp_scale / self.loss_scale
        self._build_opt(use_gscale=True)

    def calibrate_dp_schedule(self, steps=10, warm=2, log=True):
        """Chooses between the overlapped and the serialised gradient all-reduce BY MEASUREMENT on this node: ``warm`` +
        ``steps`` training steps under each schedule on whatever the input buffers hold, MAX over the ranks, the faster one stays -
        so the first multi-GPU run cannot regress on the question no 1-GPU box can answer (do RCCL's channel workgroups co-reside with
        the 100-150 KB-LDS convolution workgroups of the backward, or queue behind them?).  Every piece of mutable state (parameters,
        moments, BatchNormalization statistics, step counters) is restored afterwards, the replicas end bit-identical to how they
        started, and both schedules produce bit-identical gradients anyway (tests/test_dp_gpu.py).  Returns the record it logs."""
        import sys
        import time
        if self.reducer is None or not self.reducer.active or not self.plan.bwd_monotone:
            return None
        saved = [t.clone() for t in self._mutable_state()]
        want = self.dp_overlap
        times = {}
        two_phase = distributed.two_phase_bounds(self.plan.bwd_marks, self.plan.G.numel())
        for mode in (True, False):
            self.dp_overlap, self._segments, self._graphs, self._works = mode, None, None, []
            # (each schedule with ITS bucket bounds: two ranges for the overlap, the uniform buckets for the serialised pass - advisor, round 5)
            self.reducer.set_bounds(two_phase) if mode else self.reducer.reset_bounds()
            for _ in range(warm):
                self.forward_backward(); self.apply_gradients()
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.forward_backward(); self.apply_gradients()
            torch.cuda.synchronize(self.device)
            times[mode] = (time.perf_counter() - t0) / steps
        t = torch.tensor([times[True], times[False]], dtype=torch.float64, device=self.device)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)          # every rank takes the same decision
        t_ov, t_ser = float(t[0].item()), float(t[1].item())
        choice = bool(t_ov <= t_ser)
        self.dp_overlap, self._segments, self._graphs, self._works = choice, None, None, []
        self.reducer.set_bounds(two_phase) if choice else self.reducer.reset_bounds()
        for d, sv in zip(self._mutable_state(), saved):
            d.copy_(sv)
        torch.cuda.synchronize(self.device)
        rec = {"overlapped_ms": round(1e3 * t_ov, 4), "serialised_ms": round(1e3 * t_ser, 4), "steps": steps,
               "chosen": "overlapped" if choice else "serialised", "configured": "overlapped" if want else "serialised"}
        self.dp_schedule = rec
        if log and distributed.env_world()[0] == 0:
            print("[stp] data-parallel schedule: overlapped %.3f ms/step, serialised %.3f ms/step over %d steps -> %s "
                  "(STP_DP_OVERLAP=0|1 fixes the schedule without measuring)" % (1e3 * t_ov, 1e3 * t_ser, steps, rec["chosen"]), file=sys.stderr)
        return rec

    def _build_opt(self, use_gscale=False):
        """(Re)creates the optimizer launch list.  ``use_gscale``: multiply gradients by the device
        scalar ``gscale`` (clipnorm factor and/or 1/world_size of the data-parallel mean)."""
        p = self.plan
        p.opt = []
        n = p.P.numel()
        # clipnorm, and - whenever the loss is scaled (fp16) - the overflow guard: a non-finite gradient norm turns the step into a no-op
        # (stp_grad_global_scale writes the skip marker; parameters, moments and the step counter stay as they were)
        guard = self.loss_scale != 1.0
        if self.dls is not None:
            p._emit(p.opt, "stp_grad_global_scale_dls", p.G.data_ptr(), n, self.clipnorm, getattr(self, "dp_scale", 1.0) / self.loss_scale,
                    self.gscale.data_ptr(), self.dls.data_ptr(), self.ws_norm.data_ptr(), self.ws_norm.numel() * 4)
        elif self.clipnorm > 0 or guard:
            p._emit(p.opt, "stp_grad_global_scale", p.G.data_ptr(), n, self.clipnorm, getattr(self, "dp_scale", 1.0) / self.loss_scale,
                    self.gscale.data_ptr(), self.ws_norm.data_ptr(), self.ws_norm.numel() * 4)
        gs = self.gscale.data_ptr() if (use_gscale or self.clipnorm > 0 or guard) else None
        has_frozen = any(not i.trainable for i in p.params.values())
        mask = p.mask.data_ptr() if has_frozen else None
        kw = self.opt_kwargs
        if self.optimizer == "adam":
            p._emit(p.opt, "stp_adam", p.P.data_ptr(), p.G.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), n,
                    self.lr.data_ptr(), float(kw.get("beta_1", 0.9)), float(kw.get("beta_2", 0.999)),
                    float(kw.get("epsilon", 1e-7)), self.opt_state.data_ptr(), mask, gs, self.clipvalue)
        elif self.optimizer == "nadam":
            p._emit(p.opt, "stp_nadam", p.P.data_ptr(), p.G.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), n,
                    self.lr.data_ptr(), float(kw.get("beta_1", 0.9)), float(kw.get("beta_2", 0.999)),
                    float(kw.get("epsilon", 1e-7)), float(kw.get("schedule_decay", 0.004)), self.opt_state.data_ptr(),
                    self.opt_fstate.data_ptr(), mask, gs, self.clipvalue)
        elif self.optimizer == "rmsprop":
            p._emit(p.opt, "stp_rmsprop", p.P.data_ptr(), p.G.data_ptr(), self.m.data_ptr(), n, self.lr.data_ptr(),
                    float(kw.get("rho", 0.9)), float(kw.get("epsilon", 1e-7)), mask, gs, self.clipvalue)
        else:
            p._emit(p.opt, "stp_sgd", p.P.data_ptr(), p.G.data_ptr(), self.vel.data_ptr() if self.vel is not None else None, n,
                    self.lr.data_ptr(), float(kw.get("momentum", 0.0)), int(bool(kw.get("nesterov", False))), mask, gs,
                    self.clipvalue)
        self._graphs = None

    def _mutable_state(self):
        p = self.plan
        return [t for t in (p.P, p.S, self.opt_state, self.opt_fstate, self.m, self.v, self.vel, self.gscale, p.step_state, self.dls) if t is not None]

    # ------------------------------------------------------------------ weights
    def init_weights(self, seed=42):
        """he_uniform encoder kernels, glorot_uniform decoder/head kernels, BN gamma=1 beta=0 (the
        initialisers of classification_models / Keras Conv2D defaults)."""
        rng = np.random.RandomState(seed)
        w = OrderedDict()
        for name, info in self.plan.params.items():
            if info.kind in ("kernel", "tkernel"):
                co, kh, kw, ci = info.shape
                enc = any(name.startswith(pfx) for pfx in nets.ENCODER_PREFIXES)
                limit = np.sqrt(6.0 / (kh * kw * ci)) if enc else np.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
                w[name] = rng.uniform(-limit, limit, size=(kh, kw, ci, co) if info.kind == "kernel" else (kh, kw, co, ci)).astype(np.float32)
            elif info.kind == "dw":
                k, _, cch = info.shape
                lim = np.sqrt(6.0 / (k * k * cch + k * k))
                w[name] = rng.uniform(-lim, lim, size=info.shape + (1,)).astype(np.float32)
            elif info.kind == "gamma":
                w[name] = np.ones(info.shape, np.float32)
            else:
                w[name] = np.zeros(info.shape, np.float32)
        self.set_weights(w)
        self.plan.init_states()

    def set_weights(self, weights):
        """weights: dict name -> numpy (Keras layouts); may include BN moving statistics."""
        p = self.plan
        flat = p.P.cpu().numpy()
        st = p.S.cpu().numpy()
        for name, a in weights.items():
            a = np.asarray(a, np.float32)
            if name in p.params:
                info = p.params[name]
                if info.kind == "kernel":
                    if a.shape != (info.shape[1], info.shape[2], info.shape[3], info.shape[0]):
                        raise ValueError("%s: kernel shape %s does not match %s (HWIO)" % (name, a.shape, info.shape))
                    a = a.transpose(3, 0, 1, 2)
                elif info.kind == "dw":           # DepthwiseConv2D: Keras (kh, kw, C, 1) == the stored [kh][kw][C]
                    if a.shape not in (info.shape, info.shape + (1,)):
                        raise ValueError("%s: depthwise kernel shape %s does not match %s" % (name, a.shape, info.shape + (1,)))
                    a = a.reshape(info.shape)
                elif info.kind == "tkernel":      # Conv2DTranspose: Keras (kh, kw, out, in) -> spatially flipped OHWI
                    if a.shape != (info.shape[1], info.shape[2], info.shape[0], info.shape[3]):
                        raise ValueError("%s: kernel shape %s does not match %s (kh,kw,out,in)" % (name, a.shape, info.shape))
                    a = a[::-1, ::-1].transpose(2, 0, 1, 3)
                elif a.shape != info.shape:
                    raise ValueError("%s: shape %s does not match %s" % (name, a.shape, info.shape))
                flat[info.offset:info.offset + info.numel] = a.reshape(-1)
            elif name in p.states:
                off, numel, _ = p.states[name]
                st[off:off + numel] = a.reshape(-1)
            else:
                raise KeyError("unknown weight %r" % name)
        p.P.copy_(torch.from_numpy(flat))
        p.S.copy_(torch.from_numpy(st))

    def _unflatten(self, flat):
        out = OrderedDict()
        for name, info in self.plan.params.items():
            a = flat[info.offset:info.offset + info.numel].reshape(info.shape)
            if info.kind == "kernel":
                out[name] = a.transpose(1, 2, 3, 0).copy()
            elif info.kind == "tkernel":
                out[name] = a.transpose(1, 2, 0, 3)[::-1, ::-1].copy()
            elif info.kind == "dw":
                out[name] = a.reshape(info.shape + (1,)).copy()
            else:
                out[name] = a.copy()
        return out

    def get_weights(self):
        out = self._unflatten(self.plan.P.cpu().numpy())
        st = self.plan.S.cpu().numpy()
        for name, (off, numel, _) in self.plan.states.items():
            out[name] = st[off:off + numel].copy()
        return out

    def get_gradients(self):
        """Gradients of the last step in Keras layout (the arena holds loss_scale x gradient in fp16 mode: divided out here)."""
        m = float(self.dls[4].item()) if self.dls is not None else 1.0      # the dynamic multiplier these gradients were computed under
        return self._unflatten(self.plan.G.cpu().numpy() * np.float32(1.0 / (self.loss_scale * m)))

    @property
    def dynamic_loss_scale(self):
        """Total loss scale of the NEXT backward pass (static x dynamic multiplier); the static scale when dynamic re-scaling is off."""
        return self.loss_scale * (float(self.dls[0].item()) if self.dls is not None else 1.0)

    def broadcast_state(self, src=0):
        """Data-parallel start of a stage: every replica takes rank ``src``'s parameters, BatchNormalization moving
        statistics and optimizer state (after a checkpoint load only rank 0 read the file)."""
        distributed.broadcast_tensors(self._mutable_state(), src=src)

    def save_weights(self, path):
        """Checkpoint payload: safetensors of the Keras-layout tensors (Keras HDF5 cannot be
        produced here - h5py is absent; the file naming is the reference's, README.md:382).  Written to a temporary
        file and renamed, so a reader never sees a partial checkpoint."""
        import os
        from safetensors.numpy import save_file
        w = self.get_weights()
        tmp = "%s.tmp.%d" % (path, os.getpid())
        save_file({k: np.ascontiguousarray(v) for k, v in w.items()}, tmp,
                  metadata={"format": "stp-keras-layout", "backbone": self.backbone, "architecture": self.architecture})
        os.replace(tmp, path)

    def load_weights(self, path, strict=True):
        """Loads a checkpoint written by save_weights (``strict=False``: tensors the plan does not have are skipped -
        an encoder-only pretrained file, or a classifier head left in it).  The reference stores Keras HDF5 under the same file name
        (model.save_weights, README.md:382): such a file is recognised by its signature and rejected with a message that
        says what to do, instead of a parser error."""
        from safetensors.numpy import load_file
        with open(path, "rb") as f:
            head = f.read(8)
        if head == b"\x89HDF\r\n\x1a\n":
            raise ValueError("%s is a Keras HDF5 checkpoint written by the reference pipeline; this backend stores safetensors "
                             "with Keras-layout tensors under the same name. Export the Keras weights to a {layer/weight: array} "
                             "dict (model.get_weights with layer names) and pass it to set_weights(), then save_weights()." % path)
        w = load_file(path)
        if not strict:
            w = {k: v for k, v in w.items() if k in self.plan.params or k in self.plan.states}
        self.set_weights(w)

    # ------------------------------------------------------------------ stepping
    def _ensure_graphs(self):
        """Captures the step into two hipGraphs: 'fb' (weight copies + forward + loss + backward) and
        'opt' (optimizer); the data-parallel all-reduce runs between them.  A warm-up pass runs
        first on a snapshot of all mutable state (lazy code-object loading and
        hipFuncSetAttribute must not happen under capture), then the snapshot is restored."""
        if self._graphs is not None:
            return
        p = self.plan
        saved = [t.clone() for t in self._mutable_state()]
        p.run_prep_fwd(); p.run(p.bwd); p.run(p.opt)
        torch.cuda.synchronize()
        for t, s in zip(self._mutable_state(), saved):
            t.copy_(s)
        torch.cuda.synchronize()
        gopt = torch.cuda.CUDAGraph()
        segs = self._dp_segments()
        if segs is None:
            gfb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gfb, capture_error_mode=CAPTURE_MODE):
                p.run_prep_fwd(); p.run(p.bwd)
            self._graphs = {"fb": gfb, "opt": gopt}
        else:
            # one graph per backward segment; the first also holds the weight copies, the forward and the loss
            gs, a = [], 0
            for i, (b, _) in enumerate(segs):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    if i == 0:
                        p.run_prep_fwd()
                    p.run(p.bwd[a:b])
                gs.append(g)
                a = b
            self._graphs = {"segs": gs, "opt": gopt}
        with torch.cuda.graph(gopt, capture_error_mode=CAPTURE_MODE):
            p.run(p.opt)

    def _dp_segments(self):
        """[(end launch index, [(s, e) gradient ranges final after it])] or None when the all-reduce is not overlapped."""
        if self.reducer is None or not getattr(self, "dp_overlap", False) or not self.reducer.active:
            return None
        if self._segments is None:
            p = self.plan
            self._segments = distributed.overlap_schedule(p.bwd_marks, self.reducer.bounds(p.G.numel()), len(p.bwd))
        return self._segments

    def load_batch(self, x, y=None):
        """Copies a uint8 image batch [N,H,W,C] (and masks [N,H,W,1] in {0,1}) into the plan's input buffers."""
        p = self.plan
        xi = p.inputs["image"].buf
        x = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
        if x.dtype != torch.uint8:
            raise TypeError("images must be uint8 (raw 0..255), as SimplePNGMaskDataSet delivers them")
        xi.copy_(x.reshape(xi.shape), non_blocking=True)
        if y is not None:
            yi = p.inputs["mask"].buf
            y = torch.from_numpy(np.ascontiguousarray(y)) if isinstance(y, np.ndarray) else y
            yi.copy_(y.to(torch.uint8).reshape(yi.shape), non_blocking=True)

    def forward_backward(self):
        p = self.plan
        segs = self._dp_segments()
        if self.use_graph:
            self._ensure_graphs()
        if segs is None:
            if self.use_graph:
                self._graphs["fb"].replay()
            else:
                p.run_prep_fwd(); p.run(p.bwd)
            return
        self._works, a = [], 0
        for i, (b, ranges) in enumerate(segs):
            if self.use_graph:
                self._graphs["segs"][i].replay()
            else:
                if i == 0:
                    p.run_prep_fwd()
                p.run(p.bwd[a:b])
            a = b
            for s, e in ranges:
                self._works.append(self.reducer.allreduce_range(p.G, s, e))

    def apply_gradients(self):
        p = self.plan
        if self._dp_segments() is not None:
            for w in self._works:
                if w is not None:
                    w.wait()
            self._works = []
        elif self.reducer is not None:
            self.reducer.allreduce(p.G)
        if self.use_graph:
            self._ensure_graphs()
            self._graphs["opt"].replay()
        else:
            p.run(p.opt)

    def train_on_batch(self, x=None, y=None, fetch=True):
        """One training step.  x: uint8 [N,H,W,3], y: [N,H,W,1] in {0,1}; None = reuse the resident
        batch.  Returns dict(loss, binary_crossentropy, dice_loss, dice, binary_accuracy) if fetch."""
        if x is not None:
            self.load_batch(x, y)
        self.forward_backward()
        self.apply_gradients()
        return self.metrics() if fetch else None

    @property
    def skipped_steps(self):
        """Optimizer steps skipped by the overflow guard (non-finite gradients under loss scaling) since the model was built."""
        return int(self.gscale[1].item())

    def metrics(self):
        s = self.plan.loss_scalars.cpu().numpy()
        out = dict(zip(SCALAR_NAMES, (float(v) for v in s[:5])))
        if self.classes > 1:
            out["categorical_crossentropy"] = out.pop("binary_crossentropy")
        out["iou"], out["iot"] = float(s[8]), float(s[9])
        if len(self.loss_w) > 2:
            out["iou_loss"], out["jaccard_loss"], out["focal_loss"] = 1.0 - float(s[8]), float(s[10]), float(s[11])
            out["lovasz_loss"] = float(s[12])
        return out

    def logits(self):
        ts = self.plan.tensors                     # FPN / PSPNet / DeepLab: the head output is a resized tensor named "logits"
        t = self.plan.tensor("logits" if "logits" in ts else "final_conv")      # (a resize fused into the loss is materialised here)
        return t.buf.to(torch.float32).cpu().numpy()

    def activation(self, name):
        return self.plan.tensor(name).buf.to(torch.float32).cpu().numpy()

    def set_lr(self, lr):
        self.lr.fill_(float(lr))

    def get_lr(self):
        return float(self.lr.item())

    # ------------------------------------------------------------------ inference
    def predict(self, x):
        """``model.predict``: x uint8 [B,H,W,C] (any B) -> float32 probabilities [B,H,W,classes].
        Inference-phase BatchNormalization (moving statistics), weights shared with training."""
        if self._infer is None:
            ip = graph.Plan(self.batch, self.dtype, str(self.device), training=False)
            ip.define(self._net(False), share=self.plan)
            self._infer = ip
        ip = self._infer
        x = np.ascontiguousarray(x)
        if x.dtype != np.uint8:
            raise TypeError("images must be uint8 (raw 0..255)")
        B = x.shape[0]
        out = np.empty((B, self.H, self.W, self.classes), np.float32)
        xi = ip.inputs["image"].buf
        for s in range(0, B, self.batch):
            chunk = x[s:s + self.batch]
            n = chunk.shape[0]
            if n < self.batch:
                chunk = np.concatenate([chunk, np.zeros((self.batch - n,) + chunk.shape[1:], np.uint8)], axis=0)
            xi.copy_(torch.from_numpy(chunk).reshape(xi.shape))
            ip.run(ip.prep); ip.run(ip.fwd)
            out[s:s + n] = ip.probs[:n].cpu().numpy()
        return out
