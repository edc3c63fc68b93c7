"""The generated dK / dV pass of the attention backward (tools/gen_attn_dkv.py -> csrc/attn_dkv_asm.h) executed on the CPU by
tools/cdna_emu.py and compared with a float64 restatement of the autograd of components/attention.py:54-78 (dropout mask of
tests/hashmask.py).  One workgroup = 128 keys of one (batch, head), all queries."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hashmask import ATTN_KEY_SALT, attn_keep_mask, o2_hash64  # noqa: E402
from tests.test_attn_asm_emu_cpu import bf16_bits, bf16_val, make_qkv, nerr  # noqa: E402
from tools import cdna_emu, gen_attn_dkv  # noqa: E402

D = 128


def reference(qkv_bits, do_bits, B, L, H, b, head, p, seed):
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
    dk = np.log(2.0) * (dS.T @ qt)            # w.r.t. the UNSCALED k: dS^T q / sqrt(d) = ln 2 * dS^T q~
    dv = A.T @ do
    return dk, dv, lse2, delta, sc


def run_workgroup(qkv_bits, do_bits, B, L, H, b, head, ktile, p, seed, lse2, delta, sc):
    drop = p > 0
    thr = int(p * 256.0 + 0.5)
    lines = gen_attn_dkv.gen(drop)
    mem_bytes = 1 << 23
    QKV, DOUT, WS0, WS1, DQKV = 0x1000, 0x300000, 0x500000, 0x520000, 0x540000
    Lp = L + 64
    strideb, hd2 = 3 * H * D * 2, H * D * 2
    names = dict(kptr="s[0:1]", vptr="s[2:3]", qptr="s[4:5]", doptr="s[6:7]", lseptr="s[16:17]", dltptr="s[18:19]", okptr="s[20:21]",
                 ovptr="s[22:23]", nt="s8", strideb="s9", dorowb="s10", ldsb="s11", wave="s12", thr="s13", fk="s14", fv="s15",
                 rowbase="s24", hseed="s25", keyh="v0")
    emu = cdna_emu.Emu(lines, [names] * 4, nwaves=4, lds_bytes=gen_attn_dkv.LDS_BYTES, mem_bytes=mem_bytes, strict=True)
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
    s_lo, s_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    hseed = (s_lo ^ (((s_hi << 16) | (s_hi >> 16)) & 0xFFFFFFFF) ^ ((s_hi + (s_hi << 3)) & 0xFFFFFFFF)) & 0xFFFFFFFF
    for wv, w in enumerate(wg.waves):
        k0 = ktile * 128 + wv * 32
        vals = dict(kptr=QKV + (b * L + k0) * strideb + (H * D + head * D) * 2, vptr=QKV + (b * L + k0) * strideb + (2 * H * D + head * D) * 2,
                    qptr=QKV + b * L * strideb + head * D * 2, doptr=DOUT + (b * L * H + head) * D * 2,
                    lseptr=WS0 + (b * H + head) * Lp * 4, dltptr=WS1 + (b * H + head) * Lp * 4,
                    okptr=DQKV + (b * L + k0) * strideb + (H * D + head * D) * 2, ovptr=DQKV + (b * L + k0) * strideb + (2 * H * D + head * D) * 2)
        for name, reg in (("kptr", 0), ("vptr", 2), ("qptr", 4), ("doptr", 6), ("lseptr", 16), ("dltptr", 18), ("okptr", 20), ("ovptr", 22)):
            w.s[reg] = vals[name] & 0xFFFFFFFF
            w.s[reg + 1] = vals[name] >> 32
        w.s[8], w.s[9], w.s[10], w.s[11], w.s[12], w.s[13] = L // 64, strideb, hd2, 0, wv, thr
        w.s[14] = int(np.float32(np.log(2.0) * sc).view(np.uint32))
        w.s[15] = int(np.float32(sc).view(np.uint32))
        w.s[24], w.s[25] = (b * H + head) * L, hseed
        keys = k0 + (np.arange(64) & 31)
        w.v[0] = o2_hash64((seed ^ ATTN_KEY_SALT) & 0xFFFFFFFFFFFFFFFF, (keys >> 2).astype(np.uint64)).astype(np.uint32)
        w.m0 = 0x777
    emu.run()
    for w in wg.waves:
        assert w.m0 == 0x777 and w.exec == (1 << 64) - 1 and not w.vm and not w.lgkm
    dk = np.zeros((128, D), dtype=np.float32)
    dv = np.zeros((128, D), dtype=np.float32)
    for r in range(128):
        a = DQKV + (b * L + ktile * 128 + r) * strideb + (H * D + head * D) * 2
        dk[r] = bf16_val(wg.mem[a:a + 2 * D].view(np.uint16))
        dv[r] = bf16_val(wg.mem[a + hd2:a + hd2 + 2 * D].view(np.uint16))
    return dk, dv, emu


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_generated_dkv_matches_float64(p):
    B, L, H = 1, 512, 2
    qkv = make_qkv(B, L, H, 21)
    do = bf16_bits(np.random.default_rng(22).standard_normal((B, L, H, D)).astype(np.float32))
    seed = 0x1234567812345
    rdk, rdv, lse2, delta, sc = reference(qkv, do, B, L, H, 0, 1, p, seed)
    dk, dv, emu = run_workgroup(qkv, do, B, L, H, 0, 1, 2, p, seed, lse2, delta, sc)
    assert nerr(dv, rdv[256:384]) < 1.5e-2
    assert nerr(dk, rdk[256:384]) < 1.5e-2
    assert emu.wg.waves[0].counts["v_mfma_f32_32x32x16_bf16"] == (L // 64) * 64 + 32


def test_generated_dkv_batch_offsets():
    B, L, H = 2, 256, 1
    qkv = make_qkv(B, L, H, 23)
    do = bf16_bits(np.random.default_rng(24).standard_normal((B, L, H, D)).astype(np.float32))
    rdk, rdv, lse2, delta, sc = reference(qkv, do, B, L, H, 1, 0, 0.1, 9)
    dk, dv, _ = run_workgroup(qkv, do, B, L, H, 1, 0, 1, 0.1, 9, lse2, delta, sc)
    assert nerr(dv, rdv[128:256]) < 1.5e-2 and nerr(dk, rdk[128:256]) < 1.5e-2
