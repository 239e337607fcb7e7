import sys, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV="cuda"
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
for (N,H,W,C) in [(16,512,512,16),(16,256,256,32),(16,256,256,64),(16,128,128,64),(16,64,64,128),(16,32,32,256),(16,16,16,512)]:
    rows=N*H*W
    x=torch.randn(rows,C,device=DEV).to(torch.bfloat16); dy=torch.randn(rows,C,device=DEV).to(torch.bfloat16)
    y=torch.empty_like(x); dx=torch.empty_like(x)
    m=torch.empty(C,device=DEV); r=torch.empty(C,device=DEV); g=torch.ones(C,device=DEV); b=torch.zeros(C,device=DEV)
    dg=torch.empty(C,device=DEV); db=torch.empty(C,device=DEV)
    ws=torch.empty(ops.bn_workspace_bytes(C)//4,device=DEV)
    mb=rows*C*2/1e6
    t1=timeit(lambda: ops.bn_stats(x,rows,C,1e-3,0.99,m,r,None,None,ws))
    t2=timeit(lambda: ops.bn_apply(x,y,rows,C,C,m,r,g,b,1))
    t3=timeit(lambda: ops.bn_backward(x,dy,dx,rows,C,m,r,g,b,dg,db,1,0,ws))
    print("%-22s %7.1f MB  stats %7.1f us %5.2f TB/s | apply %7.1f us %5.2f TB/s | bwd %7.1f us %5.2f TB/s" % (str((N,H,W,C)), mb, t1, mb/t1/1e6*1e6/1e6, t2, 2*mb/t2, t3, 5*mb/t3))
