import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import nets as N, step, losses
from segmentation_training_pipeline_amd.backend import HipSegModel
bb="resnet18"
P = N.init_unet_resnet(bb, seed=42)
x,y = step.synthetic_batch(2,64,64,seed=1234)
names = N.trainable_names(P)
Pt = N.to_torch(P, names)
# hook oracle bn to capture decoder_stage4_bn1 in/out grads
orig=N._bn_apply; st={}
def hook(ctx,x_,name,eps,relu):
    out=orig(ctx,x_,name,eps,relu)
    if name=="decoder_stage4_bn1":
        x_.retain_grad(); out.retain_grad(); st["x"]=x_; st["y"]=out
    return out
N._bn_apply=hook
logits,_=N.unet_resnet_forward(Pt, torch.from_numpy(x.astype(np.float32)), bb)
l=losses.composite_loss("binary_crossentropy+1.0*dice_loss", torch.from_numpy(y.astype(np.float32)), torch.sigmoid(logits)); l.backward()
N._bn_apply=orig
nh = lambda t: t.permute(0,2,3,1).detach().numpy()
xo, yo, dyo, dxo = nh(st["x"]), nh(st["y"]), nh(st["y"].grad), nh(st["x"].grad)
m = HipSegModel("Unet", bb, (64,64,3), 1, "sigmoid", batch=2, dtype="fp32", loss="binary_crossentropy+1.0*dice_loss", use_graph=False)
m.set_weights(P)
m.load_batch(x,y); m.forward_backward(); torch.cuda.synchronize()
T = m.plan.tensors
xh = T["decoder_stage4_conv1"].buf.float().cpu().numpy(); yh = T["decoder_stage4_bn1"].buf.float().cpu().numpy()
dyh = T["decoder_stage4_bn1"].grad.float().cpu().numpy(); dxh = T["decoder_stage4_conv1"].grad.float().cpu().numpy()
rel = lambda a,b: np.linalg.norm((a-b).ravel().astype(np.float64))/ (np.linalg.norm(b.ravel().astype(np.float64))+1e-30)
print("x rel", rel(xh,xo), "y rel", rel(yh,yo), "dy rel", rel(dyh,dyo), "dx rel", rel(dxh,dxo))
print("mask mismatch count", int(((yh>0)!=(yo>0)).sum()), "of", yh.size)
g_h = (dyh*(yh>0)).astype(np.float64); g_o=(dyo*(yo>0)).astype(np.float64)
print("dbeta from hip bufs fp64", g_h.sum(axis=(0,1,2))[:4]); print("dbeta oracle bufs fp64   ", g_o.sum(axis=(0,1,2))[:4])
print("hip dbeta kernel        ", m.get_gradients()["decoder_stage4_bn1/beta"][:4]); print("oracle autograd dbeta    ", Pt["decoder_stage4_bn1/beta"].grad.numpy()[:4])
d = np.abs(dyh-dyo); print("dy err max", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "dy max", np.abs(dyo).max())
print("dy err by row band: top", d[:,0].max(), d[:,1].max(), "mid", d[:,10:50].max(), "bottom", d[:,-1].max(), " left", d[:,:,0].max(), "right", d[:,:,-1].max())
np.set_printoptions(linewidth=200, precision=6)
print("hip dbeta all  ", m.get_gradients()["decoder_stage4_bn1/beta"]); print("oracle dbeta all", Pt["decoder_stage4_bn1/beta"].grad.numpy())
print("hip dgamma all ", m.get_gradients()["decoder_stage4_bn1/gamma"]); print("oracle dgamma  ", Pt["decoder_stage4_bn1/gamma"].grad.numpy())
e = np.abs(dxh-dxo); print("dx err per channel max", e.max(axis=(0,1,2))); print("dx max per ch", np.abs(dxo).max(axis=(0,1,2)))
# recompute dx in fp64 from hip buffers
mean = xh.astype(np.float64).mean(axis=(0,1,2)); var = xh.astype(np.float64).var(axis=(0,1,2)); rstd=1/np.sqrt(var+1e-3)
xhat=(xh-mean)*rstd; gg=g_h; M=xh.shape[0]*xh.shape[1]*xh.shape[2]
dx64 = rstd*(gg - gg.sum(axis=(0,1,2))/M - xhat*(gg*xhat).sum(axis=(0,1,2))/M)
print("dx hip vs fp64-from-hip-bufs rel", rel(dxh, dx64), " oracle vs fp64 rel", rel(dxo, dx64))
