"""Times the attention kernels at the interm_1b shape (B x 24 heads x L=8192 x d=128), with and without dropout."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

def t(f, n=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for (H, L, d) in [(24, 8192, 128), (16, 512 * 8, 64), (32, 8192, 256)]:
    qkv = torch.randn(B, L, 3 * H * d, device="cuda").to(torch.bfloat16)
    do = torch.randn(B, L, H * d, device="cuda").to(torch.bfloat16)
    for p in (0.0, 0.1):
        out, lse = _hip.attn_fwd(qkv, B, L, H, d, p, 1)
        f = t(lambda: _hip.attn_fwd(qkv, B, L, H, d, p, 1))
        b = t(lambda: _hip.attn_bwd(qkv, out, do, lse, B, L, H, d, p, 1))
        fl = 4.0 * B * H * L * L * d
        print("B=%d H=%d L=%d d=%d p=%.1f | fwd %7.3f ms %5.0f TF | bwd %7.3f ms %5.0f TF (algorithmic 2x fwd; executed 3.5x: %5.0f TF)"
              % (B, H, L, d, p, f, fl / f / 1e9, b, 2 * fl / b / 1e9, 3.5 * fl / b / 1e9), flush=True)
