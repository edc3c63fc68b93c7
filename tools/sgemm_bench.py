import sys; sys.path[:0]=['/root/repo','/root/repo/orbit-2_amd']
import torch
from climate_learn import _ops
def t(f,n=20):
    for _ in range(3): f()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
D=3072
for M,tb in ((24,False),(24,True),(115,True),(1,True)):
    A=torch.randn(M,D,device='cuda'); B=torch.randn(D,D,device='cuda')
    with torch.no_grad():
        us=t(lambda: _ops.sgemm(A,B,tb=tb))
        ref=(A.double()@(B.double().t() if tb else B.double())).float()
        err=float((_ops.sgemm(A,B,tb=tb)-ref).abs().max()/ref.abs().max())
    print('M=%d tb=%d  %.1f us  err %.2e'%(M,tb,us,err))
