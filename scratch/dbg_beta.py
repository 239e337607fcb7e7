import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import nets as N, step, losses, np_ops
from segmentation_training_pipeline_amd.backend import HipSegModel
bb="resnet18"
P = N.init_unet_resnet(bb, seed=42)
x,y = step.synthetic_batch(2,64,64,seed=1234)
Pt = N.to_torch(P, ["bn_data/beta","conv0/kernel"])
orig=N._conv; store={}
def hook(ctx,x_,name,stride=1,pad=0):
    out=orig(ctx,x_,name,stride,pad)
    if name=="conv0": out.retain_grad(); store["y"]=out
    return out
N._conv=hook
logits,_=N.unet_resnet_forward(Pt, torch.from_numpy(x.astype(np.float32)), bb)
l=losses.composite_loss("binary_crossentropy+1.0*dice_loss", torch.from_numpy(y.astype(np.float32)), torch.sigmoid(logits)); l.backward()
N._conv=orig
dY_o = store["y"].grad.permute(0,2,3,1).numpy()
m = HipSegModel("Unet", bb, (64,64,3), 1, "sigmoid", batch=2, dtype="fp32", loss="binary_crossentropy+1.0*dice_loss", use_graph=False)
m.set_weights(P)
m.load_batch(x,y); m.forward_backward(); torch.cuda.synchronize()
t = m.plan.tensors["conv0"]
dY_h = t.grad.float().cpu().numpy()
print("dY max|o|", np.abs(dY_o).max(), "max err", np.abs(dY_h-dY_o).max())
err = np.abs(dY_h-dY_o)
print("err border rows", err[:, :2].max(), err[:, -2:].max(), "interior", err[:, 4:-4, 4:-4].max())
print("sum dY hip per ch max", np.abs(dY_h.astype(np.float64).sum(axis=(0,1,2))).max(), "oracle", np.abs(dY_o.astype(np.float64).sum(axis=(0,1,2))).max())
W = P["conv0/kernel"]
def strick(dY):
    S = np_ops.conv2d_wgrad(np.ones((2,64,64,1)), dY, (7,7), 2, 3)
    return (W.astype(np.float64)*S).sum(axis=(0,1,3)), S
db_o, S_o = strick(dY_o); db_h, S_h = strick(dY_h)
print("dbeta autograd", Pt["bn_data/beta"].grad.numpy()); print("S-trick(oracle dY)", db_o); print("S-trick(hip dY) fp64", db_h)
print("hip dbeta", m.get_gradients()["bn_data/beta"])
# the input tensor seen by the stem
xin = m.plan.tensors["bn_data"].buf.float().cpu().numpy()
print("ch3 unique", np.unique(xin[...,3]))
