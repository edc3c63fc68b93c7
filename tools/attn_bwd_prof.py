"""The pre-scaled d = 128 attention backward alone, for rocprofv3 passes (tools/attn_fwd_pmc.sh with PROF=attn_bwd_prof.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
flags = int(sys.argv[3]) if len(sys.argv) > 3 else _hip.ATTN_Q_PRESCALED
H, L, d = 24, 8192, 128
torch.manual_seed(0)
qkv = torch.randn(B, L, 3, H * d, device="cuda")
qkv[:, :, 0] *= 1.4426950408889634 / d ** 0.5
qkv = qkv.reshape(B, L, 3 * H * d).to(torch.bfloat16)
do = torch.randn(B, L, H * d, device="cuda").to(torch.bfloat16)
out, lse = _hip.attn_fwd(qkv, B, L, H, d, p, 1, flags=flags)
for _ in range(20):
    _hip.attn_bwd(qkv, out, do, lse, B, L, H, d, p, 1, flags=flags)
torch.cuda.synchronize()
