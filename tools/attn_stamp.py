"""Where do the attention forward kernels spend their cycles?  Diagnostic library (tools/stamp_build.sh):
   ORBIT2_HIP_LIB=orbit-2_amd/lib/alt/stamp.so python tools/attn_stamp.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import numpy as np
import torch
from climate_learn import _hip
B, L, H, d = 4, 8192, 24, 128
for mode in ("plain8", "plain4"):
    fl = _hip.ATTN_4WAVES if mode == "plain4" else 0
    for p in (0.0, 0.1):
        qkv = (torch.randn(B, L, 3, H, d, device="cuda") * 0.5).to(torch.bfloat16)
        for _ in range(2):
            _hip.attn_fwd(qkv, B, L, H, d, p, 7, flags=fl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _hip.attn_fwd(qkv, B, L, H, d, p, 7, flags=fl); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        buf = (C.c_uint * (64 * 8))()
        _hip.lib().orbit2_debug_read_attn(buf, 64 * 8)
        dd = np.frombuffer(buf, dtype=np.uint32).reshape(64, 8).astype(np.float64)
        nt = dd[0, 7]
        print("attn fwd [%s] B=%d L=%d H=%d d=%d p=%.1f: %.3f ms %.0f TF (stamped build)" % (mode, B, L, H, d, p, ms, 4.0 * B * H * L * L * d / ms / 1e9))
        if True:
            v = dd[:, :7].mean(0) / nt
            print("   wave 0 per 64-key tile: stage-issue %.0f | QK^T %.0f | softmax%s %.0f | PV %.0f | vmcnt(0) %.0f | barrier %.0f | sum %.0f (loop/nt %.0f)"
                  % (v[0], v[1], " + dropout" if p else "", v[2], v[3], v[4], v[5], v[:6].sum(), v[6]))
