import sys; sys.path[:0]=['/root/repo','/root/repo/orbit-2_amd']
import torch
from climate_learn import _hip
def t(f,n=10):
    for _ in range(3): f()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
M,D=(int(sys.argv[1]),int(sys.argv[2])) if len(sys.argv)>2 else (65536,3072)
x=torch.randn(M,D,device='cuda').bfloat16(); dy=torch.randn(M,D,device='cuda').bfloat16(); dres=torch.randn(M,D,device='cuda').bfloat16()
g=torch.randn(D,device='cuda').bfloat16(); b=torch.randn(D,device='cuda').bfloat16()
y,mean,rstd=_hip.layernorm_fwd(x,g,b)
dg=torch.empty(D,device='cuda',dtype=torch.bfloat16); db=torch.empty(D,device='cuda',dtype=torch.bfloat16)
print('ln_bwd total ms', t(lambda: _hip.layernorm_bwd(dy,x,g,mean,rstd,dres,dg,db)))
