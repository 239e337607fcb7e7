import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import nets as N, step
from segmentation_training_pipeline_amd.backend import HipSegModel
bb="resnet18"; LOSS="binary_crossentropy+1.0*dice_loss"
P = N.init_unet_resnet(bb, seed=42)
x,y = step.synthetic_batch(2,64,64,seed=1234)
tr = step.OracleTrainer(P, backbone=bb, loss=LOSS, optimizer="adam", lr=1e-3)
o = tr.step(x.astype(np.float32), y.astype(np.float32), apply=False)
res={}
for dt in ("fp32","bf16"):
    m = HipSegModel("Unet", bb, (64,64,3), 1, "sigmoid", batch=2, dtype=dt, loss=LOSS, use_graph=False)
    m.set_weights(P); m.load_batch(x,y); m.forward_backward(); torch.cuda.synchronize()
    res[dt]=m.get_gradients()
def cos(a,b):
    a=a.ravel().astype(np.float64); b=b.ravel().astype(np.float64); return a@b/(np.linalg.norm(a)*np.linalg.norm(b)+1e-30)
for k,r in o["grads"].items():
    print("%-34s cos(fp32,oracle) %.5f  cos(bf16,oracle) %.4f  norm ratio bf16/oracle %.3f" % (k, cos(res["fp32"][k],r), cos(res["bf16"][k],r), np.linalg.norm(res["bf16"][k])/ (np.linalg.norm(r)+1e-30)))
