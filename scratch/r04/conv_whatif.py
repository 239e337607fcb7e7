"""What-if timings of single conv launches of the real step: the launch record's parameter block is copied and modified
(no BatchNormalization-backward fusion / no folded shortcut / other tiles / no accumulate).  usage: python scratch/r04/conv_whatif.py"""
import sys, copy, ctypes as C, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd.backend import HipSegModel
from segmentation_training_pipeline_amd import _lib
m = HipSegModel("Unet", "resnet34", (512, 512, 3), 1, "sigmoid", batch=16, dtype="bf16", loss="binary_crossentropy+1.0*dice_loss", use_graph=False)
rng = np.random.RandomState(0)
m.load_batch(rng.randint(0, 256, (16, 512, 512, 3)).astype(np.uint8), (rng.rand(16, 512, 512, 1) < 0.2).astype(np.uint8))
m.forward_backward(); torch.cuda.synchronize()
lib = m.plan.lib
st = torch.cuda.current_stream()

def clone(p):
    q = _lib.ConvParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
    return q

def timeit(q, reps=20):
    rc = lib.stp_conv2d(C.byref(q), st.cuda_stream)
    if rc != 0:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        lib.stp_conv2d(C.byref(q), st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

want = [("stage2_unit1_conv1", "dgrad"), ("stage3_unit1_conv1", "dgrad"), ("stage1_unit1_sc", "dgrad"), ("stage1_unit1_sc", "fwd"), ("stage2_unit1_conv1", "fwd")]
import functools
print = functools.partial(print, flush=True)
for fn, args, name, meta in m.plan.fwd + m.plan.bwd:
    if name != "stp_conv2d" or not meta or (meta["layer"], meta["pass"]) not in want:
        continue
    p = args[0]._obj
    print("==== %s %s: N %d Hv %d Wv %d C0 %d -> Cout %d  k %d s %d mode %d  Ho %d Wo %d  acc0 %d bnb %s stats %s fold %s tile %d (auto %d)"
          % (meta["layer"], meta["pass"], p.N, p.Hv, p.Wv, p.C0, p.Cout, p.KH, p.stride, p.src0_mode, p.Ho, p.Wo, p.accumulate0, bool(p.bnb_x),
             bool(p.stats_partial), bool(p.fold_src), p.tile, meta.get("tile")))
    base = timeit(clone(p))
    print("  as in the step            %8.1f us   %.1f TF" % (base, meta["flops"] / base / 1e6))
    if p.bnb_x:
        q = clone(p); q.bnb_x = None; q.stats_partial = None
        print("  no BN-backward fusion     %8.1f us" % timeit(q))
    if p.stats_partial and not p.bnb_x:
        q = clone(p); q.stats_partial = None
        print("  no statistics             %8.1f us" % timeit(q))
    if p.fold_src:
        q = clone(p); q.fold_src = None; q.fold_weight = None; q.fold_C = 0
        print("  no folded shortcut        %8.1f us" % timeit(q))
        q.bnb_x = None; q.stats_partial = None
        print("  no fold, no BN-backward   %8.1f us" % timeit(q))
    if p.accumulate0:
        q = clone(p); q.accumulate0 = 0
        print("  no accumulate             %8.1f us" % timeit(q))
    if meta["pass"] == "dgrad" and meta.get("tile", 0) in (65, 70, 71):
        for tile in (meta["tile"] + 32, meta["tile"] + 64):      # same pixel / channel tile, 3 and 4 ring stages (same partial-sum columns)
            q = clone(p); q.tile = tile
            t = timeit(q)
            print("  tile %3d                  %s" % (tile, "refused" if t is None else "%8.1f us" % t))
        for tile in (69, 101, 66, 98, 70, 102, 65):      # other pixel / channel tiles: own partial-sum buffer of the size that tile needs
            q = clone(p); q.tile = tile
            nfl = int(lib.stp_conv2d_stats_floats(C.byref(q)))
            if nfl <= 0:
                print("  tile %3d                  not available" % tile); continue
            buf = torch.empty(nfl + 1024, dtype=torch.float32, device="cuda")
            q.stats_partial = buf.data_ptr()
            t = timeit(q)
            print("  tile %3d                  %s" % (tile, "refused" if t is None else "%8.1f us" % t))
