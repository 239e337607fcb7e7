import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import nets as N, step, losses
from segmentation_training_pipeline_amd.backend import HipSegModel
bb="resnet18"
P = N.init_unet_resnet(bb, seed=42)
x,y = step.synthetic_batch(2,64,64,seed=1234)
names = N.trainable_names(P)
Pt = N.to_torch(P, names)
taps={}
logits,_=N.unet_resnet_forward(Pt, torch.from_numpy(x.astype(np.float32)), bb, taps=None)
logits.retain_grad()
l=losses.composite_loss("binary_crossentropy+1.0*dice_loss", torch.from_numpy(y.astype(np.float32)), torch.sigmoid(logits)); l.backward()
m = HipSegModel("Unet", bb, (64,64,3), 1, "sigmoid", batch=2, dtype="fp32", loss="binary_crossentropy+1.0*dice_loss", use_graph=False)
m.set_weights(P)
m.load_batch(x,y); m.forward_backward(); torch.cuda.synchronize()
dl_h = m.plan.tensors["final_conv"].grad.float().cpu().numpy()[...,0]
dl_o = logits.grad.numpy()[...,0]
rel = lambda a,b: np.linalg.norm((a-b).ravel().astype(np.float64))/ (np.linalg.norm(b.ravel().astype(np.float64))+1e-30)
print("logits rel", rel(m.logits(), logits.detach().numpy()), "dlogits rel", rel(dl_h, dl_o), "max", np.abs(dl_h-dl_o).max(), np.abs(dl_o).max())
g = m.get_gradients()
for k in names:
    print("%-34s rel %.2e" % (k, rel(g[k], Pt[k].grad.numpy())))
