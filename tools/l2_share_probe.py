"""Round 6: does a cohort of workgroups on one XCD share its panel strips in L2?  TN (weight-gradient form) problems on the 4-wave
kernel whose tiles map to known cohorts, swept over the contraction length: FETCH_SIZE per launch against the bytes an ideally
sharing cohort would fetch beyond L2 and the bytes of no sharing at all.
   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/l2_share_probe.py       (tools/l2_share.sh summarises)
cases (M x N tiles):  2 x 8 = 16 workgroups, two per XCD sharing one B strip;  32 x 8 = 256 = one round, a 4 x 8 cohort per XCD;
64 x 8 = two rounds;  96 x 8 = three rounds"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
CASES = [(2, 8), (32, 8), (64, 8), (96, 8)]
KS = [8192, 32768, 131072]
if __name__ == "__main__":
    for tm, tn in CASES:
        M, N = 256 * tm, 256 * tn
        for K in KS:
            A, B = r(K, M), r(K, N)
            o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            for _ in range(4):
                _hip.gemm(A, B, o, M, N, K, M, N, N, a_kc=False, b_kc=False, tile=260)
            torch.cuda.synchronize()
            del A, B, o
    print("done")
