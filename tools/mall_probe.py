"""rocprofv3 target for separating Infinity-Cache (MALL) hits from HBM reads in the L2-miss traffic of the step's kernels.
No counter on this box sits behind the Infinity Cache (tools/profile_round.sh lists what there is: the TCC_EA0_* fabric
counters, which count hits and misses alike), but TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ is the MEAN LATENCY of an L2 miss, and
an Infinity-Cache hit is ~350 cycles shorter than an HBM read (MI355X_MICROARCH.md: 545 vs 900 cycles unloaded).  Two
calibration streams pin the two ends under load:
    hbm  : sum of a 4 GB bf16 tensor, three passes          (every line read once per 4 GB: far beyond 256 MiB)
    mall : sum of a 96 MB fp32 tensor, twenty passes        (beyond the 32 MiB of L2, resident in the Infinity Cache)
then the Block's GEMMs (forward NT, input-gradient NN, grouped weight-gradient TN) and the attention kernels run at the
interm_1b shape.  tools/summarize_prof.py mall <csv> <out.json> turns the counters into per-kernel latencies and the
interpolated hit share.   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum -- python3 tools/mall_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

BF = torch.bfloat16
big = torch.ones(2 << 30, dtype=BF, device="cuda")           # 4 GB
for _ in range(3):
    big.sum()
del big
small = torch.ones(24 << 20, dtype=torch.float32, device="cuda")  # 96 MB
for _ in range(20):
    small.sum()
torch.cuda.synchronize()
B, L, H, d, D = 8, 8192, 24, 128, 3072
M = B * L
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).to(BF)
x, w = rnd(M, D), rnd(3 * D, D)
qkv = torch.empty(M, 3 * D, dtype=BF, device="cuda")
for _ in range(2):
    _hip.gemm(x, w, qkv, M, 3 * D, D, D, D, 3 * D, a_kc=True, b_kc=True)                 # qkv forward (NT)
w1 = rnd(4 * D, D); h = torch.empty(M, 4 * D, dtype=BF, device="cuda")
for _ in range(2):
    _hip.gemm(x, w1, h, M, 4 * D, D, D, D, 4 * D, a_kc=True, b_kc=True)                  # fc1 forward (NT)
dx = torch.empty(M, D, dtype=BF, device="cuda")
for _ in range(2):
    _hip.gemm(h, w1, dx, M, D, 4 * D, 4 * D, D, D, a_kc=True, b_kc=False)                # fc1 input gradient (NN)
dw = [torch.empty(3 * D, D, dtype=BF, device="cuda"), torch.empty(4 * D, D, dtype=BF, device="cuda")]
for _ in range(2):
    _hip.gemm_grouped([(qkv, x, dw[0], 3 * D, D, M, 3 * D, D, D, dict(a_kc=False, b_kc=False)),
                       (h, x, dw[1], 4 * D, D, M, 4 * D, D, D, dict(a_kc=False, b_kc=False))])   # weight gradients (TN)
q3 = (torch.randn(B, L, 3 * H * d, device="cuda", generator=g) * 0.7).to(BF)
do = rnd(B, L, H * d)
for _ in range(2):
    out, lse = _hip.attn_fwd(q3, B, L, H, d, 0.1, 3)
    _hip.attn_bwd(q3, out, do, lse, B, L, H, d, 0.1, 3)
torch.cuda.synchronize()
print("done")
