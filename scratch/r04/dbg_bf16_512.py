"""Where does the bf16 step leave the storage-quantised oracle?  Per-tap error (in storage ulps of each tensor's range) for
U-Net/ResNet18 and ResNet34 at 64 px and 512 px (oracle on the GPU box's CPU)."""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
os.environ["STP_UPCOLLAPSE"] = "0"
from oracle import nets as onets, step as ostep
from segmentation_training_pipeline_amd.backend import HipSegModel
TAP_MAP = [("bn_data", "bn_data"), ("conv0", "conv0"), ("relu0", "bn0"), ("pooling0", "pooling0"),
           ("stage1_unit1_relu1", "stage1_unit1_bn1"), ("stage1_unit1_out", "stage1_unit1_conv2"),
           ("stage2_unit1_out", "stage2_unit1_conv2"), ("stage3_unit1_out", "stage3_unit1_conv2"),
           ("stage4_unit1_out", "stage4_unit1_conv2"), ("relu1", "bn1"),
           ("decoder_stage0_relu2", "decoder_stage0_bn2"), ("decoder_stage2_relu2", "decoder_stage2_bn2"),
           ("decoder_stage4_relu2", "decoder_stage4_bn2")]
LOSS = "binary_crossentropy+1.0*dice_loss"
for bb, size in (("resnet18", 64), ("resnet34", 64), ("resnet18", 256), ("resnet34", 256), ("resnet34", 512)):
    n = 2
    P = onets.init_unet_resnet(bb, seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=1234)
    taps = {}
    tr = ostep.OracleTrainer(P, backbone=bb, loss=LOSS, optimizer="adam", lr=1e-3, storage="bf16")
    o = tr.step(x.astype(np.float32), y.astype(np.float32), apply=False, taps=taps)
    o32 = ostep.OracleTrainer(P, backbone=bb, loss=LOSS, optimizer="adam", lr=1e-3).step(x.astype(np.float32), y.astype(np.float32), apply=False)
    m = HipSegModel("Unet", bb, (size, size, 3), 1, "sigmoid", batch=n, dtype="bf16", loss=LOSS, optimizer="Adam", lr=1e-3, use_graph=False)
    m.set_weights(P)
    m.load_batch(x, y); m.forward_backward(); torch.cuda.synchronize()
    print("==== %s @ %d" % (bb, size))
    for oname, pname in TAP_MAP:
        if oname not in taps:
            continue
        ref = taps[oname].detach().float().numpy()
        got = m.activation(pname)[..., :ref.shape[-1]]
        rng = np.abs(ref).max(); ulp = 2.0 ** (np.floor(np.log2(rng)) - 7)
        e = np.abs(got - ref)
        print("  %-24s range %8.3f  max %6.2f ulp  mean %6.3f ulp  exact %.3f" % (pname, rng, e.max() / ulp, e.mean() / ulp, (e == 0).mean()))
    ref, got = o["logits"], m.logits()
    rng = np.abs(ref).max(); ulp = 2.0 ** (np.floor(np.log2(rng)) - 7)
    e, e32 = np.abs(got - ref), np.abs(got - o32["logits"])
    print("  logits: range %.3f  vs storage-quantised max %.2f mean %.3f ulp | vs fp32 oracle max %.2f mean %.3f ulp | oracleQ vs oracle32 mean %.3f ulp"
          % (rng, e.max() / ulp, e.mean() / ulp, e32.max() / ulp, e32.mean() / ulp, np.abs(ref - o32["logits"]).mean() / ulp))
