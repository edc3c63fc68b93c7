"""Reference measurement only: the framework's fused attention (torch SDPA -> AOTriton / CK flash on ROCm) on the
interm_1b attention shape, beside orbit2_attn_{fwd,bwd}.  The product path never calls SDPA."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch, torch.nn.functional as F
from climate_learn import _hip
B, H, L, d = 4, 24, 8192, 128
if len(sys.argv) > 1: B = int(sys.argv[1])
def t(f, n=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
qkv = torch.randn(B, L, 3 * H * d, device="cuda").to(torch.bfloat16)
do = torch.randn(B, L, H * d, device="cuda").to(torch.bfloat16)
fl = 4.0 * B * H * L * L * d
for p in (0.0, 0.1):
    out, lse = _hip.attn_fwd(qkv, B, L, H, d, p, 1)
    a_f = t(lambda: _hip.attn_fwd(qkv, B, L, H, d, p, 1))
    a_b = t(lambda: _hip.attn_bwd(qkv, out, do, lse, B, L, H, d, p, 1))
    q, k, v = (x.reshape(B, L, H, d).transpose(1, 2).contiguous().requires_grad_() for x in qkv.chunk(3, dim=-1))
    g = do.reshape(B, L, H, d).transpose(1, 2).contiguous()
    try:
        with torch.nn.attention.sdpa_kernel([torch.nn.attention.SDPBackend.FLASH_ATTENTION, torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION]):
            s_f = t(lambda: F.scaled_dot_product_attention(q, k, v, dropout_p=p))
            o = F.scaled_dot_product_attention(q, k, v, dropout_p=p)
            s_b = t(lambda: torch.autograd.grad(o, (q, k, v), g, retain_graph=True))
        print("p=%.1f  ours fwd %.2f ms %4.0f TF  bwd %.2f ms %4.0f TF | SDPA fwd %.2f ms %4.0f TF  bwd %.2f ms %4.0f TF"
              % (p, a_f, fl / a_f / 1e9, a_b, 2.5 * fl / a_b / 1e9, s_f, fl / s_f / 1e9, s_b, 2.5 * fl / s_b / 1e9), flush=True)
    except Exception as e:
        print("p=%.1f ours fwd %.2f ms bwd %.2f ms | SDPA unavailable: %s" % (p, a_f, a_b, str(e)[:200]), flush=True)
