"""One shape of the attention kernels for rocprofv3 --kernel-trace --stats (per-kernel averages)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

B, H, L, d = 4, 24, 8192, 128
if len(sys.argv) > 4: B, H, L, d = (int(a) for a in sys.argv[1:5])
qkv = torch.randn(B, L, 3 * H * d, device="cuda").to(torch.bfloat16)
do = torch.randn(B, L, H * d, device="cuda").to(torch.bfloat16)
for p in (0.0, 0.1):
    for _ in range(6):
        out, lse = _hip.attn_fwd(qkv, B, L, H, d, p, 1)
        _hip.attn_bwd(qkv, out, do, lse, B, L, H, d, p, 1)
torch.cuda.synchronize()
