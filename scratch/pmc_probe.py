import sys, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV="cuda"
n,h,w,ci,co,k = 16,64,64,128,128,3
x = torch.randn(n,h,w,ci,device=DEV).to(torch.bfloat16); dy = torch.randn(n,h,w,co,device=DEV).to(torch.bfloat16)
dw = torch.empty(co,k,k,ci,device=DEV); wt = torch.randn(co,k,k,ci,device=DEV).to(torch.bfloat16); y = torch.empty(n,h,w,co,device=DEV,dtype=torch.bfloat16)
W = ops.wgrad_params(x, dy, dw, N=n,Hs0=h,Ws0=w,Hv=h,Wv=w,C0=ci,KH=k,KW=k,stride=1,pad=1,Ho=h,Wo=w,Cout=co,dtype=ops.BF16)
ws = torch.empty(ops.wgrad_workspace_bytes(W)//4+4, dtype=torch.float32, device=DEV)
P = ops.conv_params(x, wt, y, N=n,Hs0=h,Ws0=w,Hv=h,Wv=w,C0=ci,KH=k,KW=k,stride=1,pad=1,Ho=h,Wo=w,Cout=co,dtype=ops.BF16,tile=65)
for _ in range(3):
    ops.conv2d_wgrad_partial(W, ws, 2); ops.conv2d_wgrad_partial(W, ws, 1); ops.conv2d(P)
torch.cuda.synchronize()
