"""The generated dQ pass of the attention backward (tools/gen_attn_dq.py -> csrc/attn_dq_asm.h) executed on the CPU by tools/cdna_emu.py
and compared with a float64 restatement of the autograd of components/attention.py:54-78 (dropout mask of tests/hashmask.py).
One workgroup = 256 query rows of one (batch, head)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hashmask import ATTN_KEY_SALT, attn_keep_mask, o2_hash64  # noqa: E402
from tests.test_attn_asm_emu_cpu import bf16_bits, bf16_val, make_qkv, nerr  # noqa: E402
from tools import cdna_emu, gen_attn_dq  # noqa: E402

D = 128


def reference(qkv_bits, do_bits, B, L, H, b, head, p, seed):
    """float64: (dq [L, D] w.r.t. the UNSCALED q, lse2 [L] in the exp2 domain, delta [L])"""
    x = bf16_val(qkv_bits.astype(np.uint32)).reshape(B, L, 3, H, D).astype(np.float64)
    qt, k, v = x[b, :, 0, head], x[b, :, 1, head], x[b, :, 2, head]
    do = bf16_val(do_bits.astype(np.uint32)).reshape(B, L, H, D).astype(np.float64)[b, :, head]
    s2 = qt @ k.T
    m = s2.max(-1, keepdims=True)
    lse2 = (m + np.log2(np.exp2(s2 - m).sum(-1, keepdims=True)))[:, 0]
    P = np.exp2(s2 - lse2[:, None])
    M, sc = (attn_keep_mask(seed, B * H, L, p)[0][b * H + head].astype(np.float64), 256.0 / (256.0 - int(p * 256 + 0.5))) if p > 0 \
        else (np.ones((L, L)), 1.0)
    A = P * M * sc
    O = A @ v
    dA = do @ v.T
    delta = (do * O).sum(-1)
    dS = P * (dA * M * sc - delta[:, None])
    return dS @ k / np.sqrt(D), lse2, delta, sc


def run_workgroup(qkv_bits, do_bits, B, L, H, b, head, qtile, p, seed, lse2, delta, sc):
    drop = p > 0
    thr = int(p * 256.0 + 0.5)
    lines = gen_attn_dq.gen(drop)
    mem_bytes = 1 << 23
    QKV, DOUT, WS0, WS1, DQ = 0x1000, 0x300000, 0x500000, 0x520000, 0x540000
    Lp = L + 64
    strideb = 3 * H * D * 2
    binds = []
    for wv in range(4):
        q0 = qtile * 256 + wv * 64
        vals = dict(kptr=QKV + b * L * strideb + (H * D + head * D) * 2, qptr=QKV + (b * L + q0) * strideb + head * D * 2,
                    doptr=DOUT + ((b * L + q0) * H + head) * D * 2, optr=DQ + (b * L + q0) * strideb + head * D * 2,
                    lseptr=WS0 + ((b * H + head) * Lp + q0) * 4, dltptr=WS1 + ((b * H + head) * Lp + q0) * 4, q0=q0)
        binds.append(dict(kptr="s[0:1]", qptr="s[2:3]", optr="s[4:5]", doptr="s[6:7]", lseptr="s[16:17]", dltptr="s[18:19]", nt="s8",
                          strideb="s9", hd2="s10", ldsb="s11", wave="s12", thr="s13", fs="s14", dorowb="s15", rhx="v0", rhy="v1",
                          _vals=vals))
    emu = cdna_emu.Emu(lines, [{k: v for k, v in bd.items() if k != "_vals"} for bd in binds], nwaves=4,
                       lds_bytes=gen_attn_dq.KH_OFF + L + 64, mem_bytes=mem_bytes, strict=True)
    wg = emu.wg
    raw = np.ascontiguousarray(qkv_bits).view(np.uint8).reshape(-1)
    wg.mem[QKV:QKV + raw.size] = raw
    raw = np.ascontiguousarray(do_bits).view(np.uint8).reshape(-1)
    wg.mem[DOUT:DOUT + raw.size] = raw
    t0 = np.full((B * H, Lp), -1e30, dtype=np.float32)
    t1 = np.zeros((B * H, Lp), dtype=np.float32)
    t0[b * H + head, :L] = -lse2
    t1[b * H + head, :L] = -delta / sc
    wg.mem[WS0:WS0 + t0.nbytes] = t0.view(np.uint8).reshape(-1)
    wg.mem[WS1:WS1 + t1.nbytes] = t1.view(np.uint8).reshape(-1)
    if drop:
        idx = np.arange(L // 4 + 16, dtype=np.uint64)
        T, wq = idx >> np.uint64(4), idx & np.uint64(15)
        kg = T * np.uint64(16) + np.uint64(2) * (wq & np.uint64(7)) + (wq >> np.uint64(3))
        tab = o2_hash64((seed ^ ATTN_KEY_SALT) & 0xFFFFFFFFFFFFFFFF, kg).astype(np.uint32)
        wg.lds[gen_attn_dq.KH_OFF:gen_attn_dq.KH_OFF + 4 * tab.size] = tab.view(np.uint8)
    fs = np.float32(sc / np.sqrt(D))
    for wv, w in enumerate(wg.waves):
        vals = binds[wv]["_vals"]
        for name, reg in (("kptr", 0), ("qptr", 2), ("optr", 4), ("doptr", 6), ("lseptr", 16), ("dltptr", 18)):
            w.s[reg] = vals[name] & 0xFFFFFFFF
            w.s[reg + 1] = vals[name] >> 32
        w.s[8], w.s[9], w.s[10], w.s[11], w.s[12], w.s[13] = L // 64, strideb, H * D * 2, 0, wv, thr
        w.s[14] = int(fs.view(np.uint32))
        w.s[15] = H * D * 2
        rows = (b * H + head) * L + vals["q0"] + (np.arange(64) & 31)
        w.v[0] = o2_hash64(seed, rows.astype(np.uint64)).astype(np.uint32)
        w.v[1] = o2_hash64(seed, (rows + 32).astype(np.uint64)).astype(np.uint32)
        w.m0 = 0x4321
    emu.run()
    for w in wg.waves:
        assert w.m0 == 0x4321 and w.exec == (1 << 64) - 1 and not w.vm and not w.lgkm
    out = np.zeros((256, D), dtype=np.float32)
    for r in range(256):
        a = DQ + (b * L + qtile * 256 + r) * strideb + head * D * 2
        out[r] = bf16_val(wg.mem[a:a + 2 * D].view(np.uint16))
    return out, emu


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_generated_dq_matches_float64(p):
    B, L, H = 1, 512, 2
    qkv = make_qkv(B, L, H, 11)
    do = bf16_bits(np.random.default_rng(12).standard_normal((B, L, H, D)).astype(np.float32))
    ref, lse2, delta, sc = reference(qkv, do, B, L, H, 0, 1, p, 777)
    out, emu = run_workgroup(qkv, do, B, L, H, 0, 1, 1, p, 777, lse2, delta, sc)
    assert nerr(out, ref[256:512]) < 1.5e-2
    assert emu.wg.waves[0].counts["v_mfma_f32_32x32x16_bf16"] == (L // 64) * 96 + 16


def test_generated_dq_batch_offsets():
    B, L, H = 2, 256, 1
    qkv = make_qkv(B, L, H, 13)
    do = bf16_bits(np.random.default_rng(14).standard_normal((B, L, H, D)).astype(np.float32))
    ref, lse2, delta, sc = reference(qkv, do, B, L, H, 1, 0, 0.1, 5)
    out, _ = run_workgroup(qkv, do, B, L, H, 1, 0, 0, 0.1, 5, lse2, delta, sc)
    assert nerr(out, ref[:256]) < 1.5e-2
