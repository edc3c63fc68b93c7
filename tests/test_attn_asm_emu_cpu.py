"""The generated attention forward (tools/gen_attn_fwd.py -> csrc/attn_fwd_asm.h) executed instruction by instruction on the CPU
(tools/cdna_emu.py: register files, LDS, LDS-DMA, MFMA lane maps, transposing LDS reads, counted waits and the software wait states the
hardware does not interlock) and compared with a numpy restatement of components/attention.py:54-78 with the kernels' own dropout
mask (tests/hashmask.py).  One workgroup = 256 query rows of one (batch, head)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hashmask import ATTN_KEY_SALT, attn_keep_mask, o2_hash64  # noqa: E402
from tools import cdna_emu, gen_attn_fwd  # noqa: E402

D = 128


def bf16_bits(x):
    return cdna_emu.bf16_round(np.asarray(x, dtype=np.float32)).astype(np.uint16)


def bf16_val(b):
    return cdna_emu.bf16_to_f32(np.asarray(b, dtype=np.uint32))


def run_workgroup(qkv_bits, B, L, H, b, head, qtile, p, seed, strict=True, cfg=None):
    """qkv_bits: uint16 [B, L, 3, H, D] (q pre-scaled).  Returns out [256, D] float, lse [256]."""
    drop = p > 0
    thr = int(p * 256.0 + 0.5)
    lines = gen_attn_fwd.gen(drop, dict(gen_attn_fwd.BASE, **(cfg or {})))
    mem_bytes = 1 << 23
    QKV, OUT, LSE = 0x1000, 0x400000, 0x600000
    strideb = 3 * H * D * 2
    binds = []
    for wv in range(4):
        q0 = qtile * 256 + wv * 64
        kptr = QKV + b * L * strideb + (H * D + head * D) * 2
        qptr = QKV + (b * L + q0) * strideb + head * D * 2
        optr = OUT + ((b * L + q0) * H + head) * D * 2
        lptr = LSE + ((b * H + head) * L + q0) * 4
        binds.append(dict(kptr="s[0:1]", qptr="s[2:3]", optr="s[4:5]", lptr="s[6:7]", nt="s8", strideb="s9", hd2="s10", ldsb="s11",
                          wave="s12", thr="s13", dscale="s14", orowb="s15", rhx="v0", rhy="v1",
                          _vals=dict(kptr=kptr, qptr=qptr, optr=optr, lptr=lptr, q0=q0)))
    emu = cdna_emu.Emu(lines, [{k: v for k, v in bd.items() if k != "_vals"} for bd in binds], nwaves=4, lds_bytes=65536 + L + 64,
                       mem_bytes=mem_bytes, strict=strict)
    wg = emu.wg
    raw = np.ascontiguousarray(qkv_bits).view(np.uint8).reshape(-1)
    wg.mem[QKV:QKV + raw.size] = raw
    # key-group hash table (csrc/attn.hip fills it in the kernel's prologue): position T*16 + h*8 + j holds K(T*16 + 2 j + h)
    if drop:
        idx = np.arange(L // 4 + 16, dtype=np.uint64)
        T, wq = idx >> np.uint64(4), idx & np.uint64(15)
        kg = T * np.uint64(16) + np.uint64(2) * (wq & np.uint64(7)) + (wq >> np.uint64(3))
        tab = o2_hash64((seed ^ ATTN_KEY_SALT) & 0xFFFFFFFFFFFFFFFF, kg).astype(np.uint32)
        wg.lds[gen_attn_fwd.KH_OFF:gen_attn_fwd.KH_OFF + 4 * tab.size] = tab.view(np.uint8)
    dscale = 256.0 / (256.0 - thr)
    for wv, w in enumerate(wg.waves):
        vals = binds[wv]["_vals"]
        for name, reg in (("kptr", 0), ("qptr", 2), ("optr", 4), ("lptr", 6)):
            w.s[reg] = vals[name] & 0xFFFFFFFF
            w.s[reg + 1] = vals[name] >> 32
        w.s[8], w.s[9], w.s[10], w.s[11], w.s[12], w.s[13] = L // 64, strideb, H * D * 2, 0, wv, thr
        w.s[14] = int(np.float32(dscale).view(np.uint32))
        w.s[15] = H * D * 2
        rows = (b * H + head) * L + vals["q0"] + (np.arange(64) & 31)
        w.v[0] = o2_hash64(seed, rows.astype(np.uint64)).astype(np.uint32)
        w.v[1] = o2_hash64(seed, (rows + 32).astype(np.uint64)).astype(np.uint32)
        w.m0 = 0x1234
    emu.run()
    for w in wg.waves:
        assert w.m0 == 0x1234 and w.exec == (1 << 64) - 1
        assert not w.vm and not w.lgkm
    out = np.zeros((256, D), dtype=np.float32)
    lse = np.zeros(256, dtype=np.float32)
    for r in range(256):
        a = OUT + ((b * L + qtile * 256 + r) * H + head) * D * 2
        out[r] = bf16_val(wg.mem[a:a + 2 * D].view(np.uint16))
        a = LSE + ((b * H + head) * L + qtile * 256 + r) * 4
        lse[r] = wg.mem[a:a + 4].view(np.float32)[0]
    return out, lse, emu


def reference(qkv_bits, B, L, H, b, head, qtile, p, seed):
    x = bf16_val(qkv_bits.astype(np.uint32)).reshape(B, L, 3, H, D).astype(np.float64)
    q = x[b, qtile * 256:(qtile + 1) * 256, 0, head]            # pre-scaled: scores are exp2 arguments
    k, v = x[b, :, 1, head], x[b, :, 2, head]
    s = q @ k.T
    m = s.max(-1, keepdims=True)
    pr = np.exp2(s - m)
    l = pr.sum(-1, keepdims=True)
    lse = (m + np.log2(l))[:, 0] * np.log(2.0)
    a = pr / l
    if p > 0:
        mask, sc = attn_keep_mask(seed, B * H, L, p)
        a = a * mask[b * H + head, qtile * 256:(qtile + 1) * 256] * sc
    return a @ v, lse


def make_qkv(B, L, H, seed, qscale=1.0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, L, 3, H, D)).astype(np.float32)
    x[:, :, 0] *= qscale * 1.4426950408889634 / np.sqrt(D)     # what the qkv GEMM's column scale stores
    return bf16_bits(x)


def nerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_generated_forward_matches_numpy(p):
    B, L, H = 1, 256, 2
    qkv = make_qkv(B, L, H, 3)
    out, lse, emu = run_workgroup(qkv, B, L, H, 0, 1, 0, p, 0x1234567)
    ref, lse_ref = reference(qkv, B, L, H, 0, 1, 0, p, 0x1234567)
    assert nerr(out, ref) < 1e-2
    assert np.abs(lse - lse_ref).max() < 2e-3
    w = emu.wg.waves[0]
    assert w.counts["v_mfma_f32_32x32x16_bf16"] == (L // 64) * 64 + 16 + 48     # loop + tail + prologue (reference, S_X(0))


def test_generated_forward_second_tile_and_batch():
    """query tile 1 of batch 1 (pointer arithmetic of Q / out / lse, row hashes), 512 keys"""
    B, L, H = 2, 512, 1
    qkv = make_qkv(B, L, H, 5)
    out, lse, _ = run_workgroup(qkv, B, L, H, 1, 0, 1, 0.1, 99)
    ref, lse_ref = reference(qkv, B, L, H, 1, 0, 1, 0.1, 99)
    assert nerr(out, ref) < 1e-2
    assert np.abs(lse - lse_ref).max() < 2e-3


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_generated_forward_late_maximum_takes_the_fixup(p):
    """rows whose scores outgrow the first tile's maximum by far more than 2^40: the guard branches to the fix-up (reference
    moved, O / l rescaled, the tile redone) -- exact result either way"""
    B, L, H = 1, 512, 1
    qkv = make_qkv(B, L, H, 7)
    x = bf16_val(qkv.astype(np.uint32)).reshape(B, L, 3, H, D)
    # key 300 (tile 4) aligned with query rows 5 and 77 of the tile, key 450 with row 200: scores of +60 .. +130 bits
    for qrow, key, mult in ((5, 300, 6.0), (77, 300, 9.0), (200, 450, 14.0)):
        x[0, key, 1, 0] = x[0, qrow, 0, 0] * mult / max(1e-6, float(np.abs(x[0, qrow, 0, 0]).max())) * 3.0
    qkv = bf16_bits(x)
    out, lse, emu = run_workgroup(qkv, B, L, H, 0, 0, 0, p, 4242)
    ref, lse_ref = reference(qkv, B, L, H, 0, 0, 0, p, 4242)
    s = (bf16_val(qkv.astype(np.uint32)).reshape(B, L, 3, H, D)[0, :256, 0, 0].astype(np.float64)
         @ bf16_val(qkv.astype(np.uint32)).reshape(B, L, 3, H, D)[0, :, 1, 0].astype(np.float64).T)
    assert (s[:, 64:].max(-1) - s[:, :64].max(-1)).max() > 45.0          # the test really forces the branch
    assert sum(w.counts.get("v_accvgpr_write_b32", 0) for w in emu.wg.waves) > 4 * 128       # a fix-up ran (O rescaled)
    assert nerr(out, ref) < 1e-2
    assert np.abs(lse - lse_ref).max() < 2e-3 * max(1.0, float(np.abs(lse_ref).max()))
