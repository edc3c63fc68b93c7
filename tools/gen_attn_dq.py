#!/usr/bin/env python3
"""Writes orbit-2_amd/csrc/attn_dq_asm.h: the whole body of attn_bwd_dq_w4_kernel (csrc/attn.hip) -- the dQ pass of the attention
backward, d = 128, q stored pre-scaled, L % 256 == 0 -- as ONE `asm volatile` statement placed by this script, on the construction
of tools/gen_attn_fwd.py (read that header first: lane maps, LDS image, LDS-DMA ring, the emulator that runs the same text).

Math (autograd of components/attention.py:54-78; csrc/attn.hip attn_bwd_dq_kernel is the compiler-scheduled form):
  S^T = K Q~^T - lse2          (accumulators start at -lse log2 e: p = exp2(S^T))
  dP^T = V dO^T                (dropped elements -> 0), t = dP^T - delta / dscale
  dS^T = p * t                 -> bf16 pairs -> dQ^T += K^T dS^T;   dq = dQ * (dscale / sqrt(d))
Shape: workgroup = 4 waves = 256 query rows, a wave = two 32-row blocks X, Y, one wave per SIMD.
  * a UNIT = (block, 32-key half kb of a 64-key tile): 24 MFMAs = S chain (8) + dP chain (8) + dQ^T += K^T(prev half) dS^T(prev) (8).
    Units alternate X, Y; the vector work of a unit (16 score elements per lane: exp2, mask, add, multiply, pack) runs in the gaps
    of the OTHER block's next unit; its dS^T feeds the dQ MFMAs of its own block's next unit.  Per tile t, phases
      A0: X(t,0) + dQ_X(t-1,1) | A1: Y(t,0) + dQ_Y(t-1,1) | A2: X(t,1) + dQ_X(t,0) | A3: Y(t,1) + dQ_Y(t,0)
  * registers: a[0:63] / a[64:127] dQ^T of X / Y, a[128:159] / a[160:191] Q fragments, a[192:223] / a[224:255] dO fragments;
    v[32:47] S^T and v[48:63] dP^T of X (one key half), v[64:95] the same of Y, v[96:111] / v[112:127] -lse2 of X / Y,
    v[128:135] / v[136:143] dS^T pairs, v[144:175] K row fragments, v[176:207] V row fragments, v[208:239] K^T fragments of one
    key half: every fragment is read from LDS once and feeds X's and Y's unit, re-filled in place in Y's phase.
  * LDS: 4-slot ring, K tiles at slot * 16 KiB, V tiles at 64 KiB + slot * 16 KiB (every K read within one 16-bit immediate of
    the lane's base), key-group hash table behind; K(t+2), V(t+2) are put in flight behind the one barrier of tile t.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_attn_fwd as F  # noqa: E402
from gen_attn_fwd import A, S, V  # noqa: E402

DQ = {"X": 0, "Y": 64}
Q_ = {"X": 128, "Y": 160}
DO = {"X": 192, "Y": 224}
S_ = {"X": 32, "Y": 64}
DP = {"X": 48, "Y": 80}
NL = {"X": 96, "Y": 112}
DS = {"X": 128, "Y": 136}
KR, VR, KT = 144, 176, 208
KH = [240, 8]
HH4 = [16, 17, 18, 19]
NDL = {"X": 20, "Y": 21}
RH = {"X": 22, "Y": 23}
VKE, VKO, VT1, VT2, VDE, VDO, VKH, VLANE = 24, 25, 26, 27, 28, 29, 30, 31
VVE, VVO, VTHR, HT, T0, T1, T2, T3 = 248, 249, 250, 251, 252, 253, 254, 255
S_DK, S_DV, S_PC, S_LW, S_T, S_TB, S_NT1, S_OFK, S_OFV, S_TMP, S_M0, S_MIX = 36, 40, 44, 48, 49, 50, 51, 52, 53, 54, 56, 57
S_QP, S_OP, S_X, S_Y2, S_EX, S_LDS, S_DP2 = 60, 62, 64, 65, 66, 68, 70

KH_OFF = 131072           # 4 K slots + 4 V slots of 16 KiB


def mfma(d, a, b, c):
    return "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (d, a, b, c)


def unit_mfmas(blk):
    """24 MFMAs of a unit: the S and dP chains alternate in slots 0..15 (K / V row fragment i in slots 2 i / 2 i + 1), the dQ
    product follows in slots 16..23 (K^T fragment i in slot 16 + i) -- the unit's S^T / dP^T are 8 MFMAs old when the next
    phase's vector stream reads them (an MFMA result has no hardware interlock against a vector read)"""
    out = []
    s, dp = V(S_[blk], 16), V(DP[blk], 16)
    for i in range(8):
        out.append(mfma(s, V(KR + 4 * i, 4), A(Q_[blk] + 4 * i, 4), V(NL[blk], 16) if i == 0 else s))
        out.append(mfma(dp, V(VR + 4 * i, 4), A(DO[blk] + 4 * i, 4), "0" if i == 0 else dp))
    return out + dq_only(blk)


def dq_only(blk):
    out = []
    for i in range(8):
        ss, db = i >> 2, i & 3
        d = A(DQ[blk] + 16 * db, 16)
        out.append(mfma(d, V(KT + 4 * i, 4), V(DS[blk] + 4 * ss, 4), d))
    return out


def kr_read(i, slot, kb):
    return "ds_read_b128 %s, %s offset:%d" % (V(KR + 4 * i, 4), V(VKO if i & 1 else VKE), slot * 16384 + kb * 8192 + 512 * (i >> 1))


def vr_read(i, slot, kb):
    return "ds_read_b128 %s, %s offset:%d" % (V(VR + 4 * i, 4), V(VVO if i & 1 else VVE), slot * 16384 + kb * 8192 + 512 * (i >> 1))


def kt_reads(i, slot, kb):
    ss, db = i >> 2, i & 3
    off = slot * 16384 + 2048 * (4 * kb + 2 * ss) + 512 * db
    return ["ds_read_b64_tr_b16 %s, %s offset:%d" % (V(KT + 4 * i, 2), V(VT1), off),
            "ds_read_b64_tr_b16 %s, %s offset:%d" % (V(KT + 4 * i + 2, 2), V(VT2), off)]


def kh_reads(par):
    return ["ds_read_b128 %s, %s" % (V(KH[par], 4), V(VKH)), "ds_read_b128 %s, %s offset:16" % (V(KH[par] + 4, 4), V(VKH)),
            "v_add_u32 %s, 64, %s" % (V(VKH), V(VKH))]


def dma_piece(which, j, slot):
    sof, desc, base = (S_OFK, S_DK, 0) if which == "K" else (S_OFV, S_DV, 65536)
    return ["s_add_u32 %s, %s, %s" % (S(S_TMP), S(sof), S(S_PC + j)),
            "s_add_u32 m0, %s, %d" % (S(S_LW), base + slot * 16384 + j * 1024),
            "s_nop 0",
            "buffer_load_dwordx4 %s, %s, %s offen lds" % (V(VDO if j >= 2 else VDE), S(desc, 4), S(S_TMP))]


def tile_offset(dt):
    return ["s_add_u32 %s, %s, %d" % (S(S_X), S(S_T), dt), "s_min_u32 %s, %s, %s" % (S(S_X), S(S_X), S(S_NT1)),
            "s_mul_i32 %s, %s, %s" % (S(S_OFK), S(S_X), S(S_TB)), "s_mov_b32 %s, %s" % (S(S_OFV), S(S_OFK))]


def hashg(blk, khreg, slot):
    hh = HH4[slot]
    return ["v_xor_b32 %s, %s, %s" % (V(hh), V(RH[blk]), V(khreg)),
            "v_mul_lo_u32 %s, %s, %s" % (V(hh), V(hh), S(S_MIX)),
            "v_xor_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" % (V(hh), V(hh), V(hh))]


def stream_gaps(blk, drop, kh, nxt):
    """the vector work of one unit of `blk` (its 16 S^T / dP^T registers -> 8 registers of dS^T pairs) as 24 per-gap lists:
    element i has its exp2 in gap 3 i // 2 together with the mask / add of its dP, the multiply in the next gap, the pack after
    the odd element's multiply.  kh = first of the unit's 4 key-group hash registers; its group 0 mix was computed by the
    previous phase; nxt = (block, register) of the NEXT unit's group 0, mixed here in the last gaps."""
    gaps = [[] for _ in range(24)]
    sb, db, ds = S_[blk], DP[blk], DS[blk]
    eg = lambda i: (3 * i) // 2
    for i in range(16):
        g = eg(i)
        gl = i >> 2
        if drop and (i & 3) == 0 and gl + 1 < 4:
            h = hashg(blk, kh + gl + 1, gl + 1)               # the next group's mix, one instruction per gap
            for k, ins in enumerate(h):
                gaps[min(23, g + k)].append(ins)
        gaps[g].append("v_exp_f32 %s, %s" % (V(sb + i), V(sb + i)))
        if drop:
            gaps[g].append("v_cmp_ge_u32_sdwa vcc, %s, %s src0_sel:BYTE_%d src1_sel:DWORD" % (V(HH4[gl]), V(VTHR), i & 3))
            gaps[g].append("v_cndmask_b32 %s, 0, %s, vcc" % (V(db + i), V(db + i)))
        gaps[g].append("v_add_f32 %s, %s, %s" % (V(db + i), V(db + i), V(NDL[blk])))
        gaps[g + 1].append("v_mul_f32 %s, %s, %s" % (V(db + i), V(sb + i), V(db + i)))
        if i & 1:
            gaps[g + 1].append("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(ds + (i >> 1)), V(db + i - 1), V(db + i)))
    if drop:
        nb, nreg = nxt
        h = hashg(nb, nreg, 0)
        gaps[22] += h[:2]
        gaps[23] += h[2:]
    return gaps


def place(mf, gaps, fixed):
    out = []
    for m, ins in enumerate(mf):
        out.append(ins)
        out += fixed.get(m, [])
        if m < len(gaps):
            out += gaps[m]
    return out


def phase(ph, q, drop, cfg):
    """phase ph (0..3 = A0..A3) of the tile in ring slot q (= t & 3).  Returns the flat instruction list."""
    blk_m = "X" if ph % 2 == 0 else "Y"
    blk_v = "Y" if ph % 2 == 0 else "X"
    mf = unit_mfmas(blk_m)
    fixed = {}

    def add(g, ins):
        fixed.setdefault(g, []).extend(ins if isinstance(ins, list) else [ins])

    # the stream's unit: A0 -> Y(t-1,1), A1 -> X(t,0), A2 -> Y(t,0), A3 -> X(t,1); kh registers by tile parity
    par = q & 1
    unit_kh = {0: KH[par ^ 1] + 4, 1: KH[par], 2: KH[par], 3: KH[par] + 4}[ph]
    nxt = {0: ("X", KH[par]), 1: ("Y", KH[par]), 2: ("X", KH[par] + 4), 3: ("Y", KH[par] + 4)}[ph]
    gaps = stream_gaps(blk_v, drop, unit_kh, nxt)
    if cfg.get("abl_valu"):
        gaps = [[] for _ in range(24)]
    if ph == 0:
        # tile t + 1 has landed for every wave (first read: A3); the slot written next held tile t - 2
        add(0, ["s_waitcnt vmcnt(0)", "s_barrier"] + tile_offset(2))
        pieces = [("K", j) for j in range(4)] + [("V", j) for j in range(4)]
        g = 2
        for w, j in pieces:
            add(g, dma_piece(w, j, (q + 2) & 3))
            g += cfg["dstride"]
        assert g - cfg["dstride"] <= 23
    if ph == 2 and drop:
        add(1, kh_reads(par ^ 1))                      # key-group hashes of tile t + 1
    if ph in (1, 3):
        # Y's unit is the second user of every fragment: re-fill in place for the next key half
        kb_n, slot_n = (1, q) if ph == 1 else (0, (q + 1) & 3)     # K / V rows: (t, 1) in A1, (t + 1, 0) in A3
        kb_t, slot_t = (0, q) if ph == 1 else (1, q)               # K^T: (t, 0) in A1, (t, 1) in A3
        for i in range(8):
            add(2 * i + cfg["lag"], kr_read(i, slot_n, kb_n))
            add(2 * i + 1 + cfg["lag"], vr_read(i, slot_n, kb_n))
            add(min(23, 16 + i + cfg["lag"]), kt_reads(i, slot_t, kb_t))
    if ph == 3:
        add(23, "s_add_u32 %s, %s, 1" % (S(S_T), S(S_T)))
    return place(mf, gaps, fixed)


def prologue(drop):
    L = []
    e = L.append
    e("s_nop 4")
    e("s_mov_b32 %s, m0" % S(S_M0))
    e("v_mbcnt_lo_u32_b32 %s, -1, 0" % V(VLANE))
    e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (V(VLANE), V(VLANE)))
    e("s_mov_b32 %s, %%[ldsb]" % S(S_LDS))
    # row reads (K and V images are the same: V's base is 64 KiB up)
    e("v_and_b32 %s, 31, %s" % (V(T0), V(VLANE)))
    e("v_lshrrev_b32 %s, 5, %s" % (V(T1), V(VLANE)))
    e("v_lshrrev_b32 %s, 3, %s" % (V(T2), V(T0)))
    e("v_lshlrev_b32 %s, 11, %s" % (V(VKE), V(T2)))
    e("v_and_b32 %s, 7, %s" % (V(T2), V(T0)))
    e("v_lshl_add_u32 %s, %s, 6, %s" % (V(VKE), V(T2), V(VKE)))
    e("v_bfe_u32 %s, %s, 2, 2" % (V(T2), V(T0)))
    e("v_xor_b32 %s, %s, %s" % (V(T2), V(T2), V(T1)))
    e("v_lshl_add_u32 %s, %s, 4, %s" % (V(VKE), V(T2), V(VKE)))
    e("v_add_u32 %s, %s, %s" % (V(VKE), S(S_LDS), V(VKE)))
    e("v_xor_b32 %s, 32, %s" % (V(VKO), V(VKE)))
    e("v_add_u32 %s, 0x10000, %s" % (V(VVE), V(VKE)))
    e("v_add_u32 %s, 0x10000, %s" % (V(VVO), V(VKO)))
    # transposed reads of K
    e("v_bfe_u32 %s, %s, 2, 2" % (V(T2), V(VLANE)))
    e("v_lshl_add_u32 %s, %s, 2, %s" % (V(T2), V(T1), V(T2)))
    e("v_lshlrev_b32 %s, 6, %s" % (V(VT1), V(T2)))
    e("v_bfe_u32 %s, %s, 4, 1" % (V(T2), V(VLANE)))
    e("v_bfe_u32 %s, %s, 1, 1" % (V(T3), V(VLANE)))
    e("v_lshl_add_u32 %s, %s, 1, %s" % (V(T2), V(T2), V(T3)))
    e("v_xor_b32 %s, %s, %s" % (V(T2), V(T2), V(T1)))
    e("v_lshl_add_u32 %s, %s, 4, %s" % (V(VT1), V(T2), V(VT1)))
    e("v_and_b32 %s, 1, %s" % (V(T2), V(VLANE)))
    e("v_lshl_add_u32 %s, %s, 3, %s" % (V(VT1), V(T2), V(VT1)))
    e("v_add_u32 %s, %s, %s" % (V(VT1), S(S_LDS), V(VT1)))
    e("v_xor_b32 %s, 32, %s" % (V(VT2), V(VT1)))
    e("v_add_u32 %s, 0x800, %s" % (V(VT2), V(VT2)))
    # LDS-DMA source offsets
    e("v_bfe_u32 %s, %s, 2, 3" % (V(T2), V(VLANE)))
    e("v_mul_lo_u32 %s, %s, %%[strideb]" % (V(VDE), V(T2)))
    e("v_bfe_u32 %s, %s, 4, 1" % (V(T2), V(VLANE)))
    e("v_and_b32 %s, 3, %s" % (V(T3), V(VLANE)))
    e("v_xor_b32 %s, %s, %s" % (V(T2), V(T2), V(T3)))
    e("v_lshl_add_u32 %s, %s, 2, %s" % (V(T2), V(T1), V(T2)))
    e("v_lshl_add_u32 %s, %s, 4, %s" % (V(VDE), V(T2), V(VDE)))
    e("v_xor_b32 %s, 32, %s" % (V(VDO), V(VDE)))
    e("v_lshlrev_b32 %s, 5, %s" % (V(VKH), V(T1)))
    e("v_add_u32 %s, %s, %s" % (V(VKH), S(S_LDS), V(VKH)))
    e("v_add_u32 %s, 0x%x, %s" % (V(VKH), KH_OFF, V(VKH)))
    e("v_mov_b32 %s, %%[rhx]" % V(RH["X"]))
    e("v_mov_b32 %s, %%[rhy]" % V(RH["Y"]))
    e("v_mov_b32 %s, %%[thr]" % V(VTHR))
    e("s_mov_b32 %s, 0x9E3779B1" % S(S_MIX))
    e("s_mov_b64 %s, %%[kptr]" % S(S_DK, 2))
    e("s_mov_b32 %s, 0x7fffffff" % S(S_DK + 2))
    e("s_mov_b32 %s, 0x00020000" % S(S_DK + 3))
    e("s_add_u32 %s, %s, %%[hd2]" % (S(S_DV), S(S_DK)))
    e("s_addc_u32 %s, %s, 0" % (S(S_DV + 1), S(S_DK + 1)))
    e("s_mov_b32 %s, 0x7fffffff" % S(S_DV + 2))
    e("s_mov_b32 %s, 0x00020000" % S(S_DV + 3))
    e("s_lshl_b32 %s, %%[strideb], 6" % S(S_TB))
    e("s_sub_u32 %s, %%[nt], 1" % S(S_NT1))
    e("s_lshl_b32 %s, %%[wave], 4" % S(S_X))
    e("s_mul_i32 %s, %s, %%[strideb]" % (S(S_PC), S(S_X)))
    e("s_add_u32 %s, %s, 128" % (S(S_PC + 1), S(S_PC)))
    e("s_lshl_b32 %s, %%[strideb], 3" % S(S_X))
    e("s_add_u32 %s, %s, %s" % (S(S_PC + 2), S(S_PC), S(S_X)))
    e("s_add_u32 %s, %s, 128" % (S(S_PC + 3), S(S_PC + 2)))
    e("s_lshl_b32 %s, %%[wave], 12" % S(S_X))
    e("s_add_u32 %s, %s, %s" % (S(S_LW), S(S_LDS), S(S_X)))
    # tiles 0, 1 -> slots 0, 1
    for tile in range(2):
        e("s_mov_b32 %s, 0" % S(S_T))
        L.extend(tile_offset(tile))
        for w in ("K", "V"):
            for j in range(4):
                L.extend(dma_piece(w, j, tile))
    # Q / dO fragments, row statistics
    e("v_lshlrev_b32 %s, 4, %s" % (V(T2), V(T1)))
    e("v_mul_lo_u32 %s, %s, %%[strideb]" % (V(T3), V(T0)))
    e("v_add_u32 %s, %s, %s" % (V(T3), V(T3), V(T2)))                 # Q: r stride + 16 h
    e("s_lshl_b32 %s, %%[strideb], 5" % S(S_X))
    e("s_mov_b64 %s, %%[qptr]" % S(S_QP, 2))
    e("s_add_u32 %s, %s, %s" % (S(S_QP), S(S_QP), S(S_X)))
    e("s_addc_u32 %s, %s, 0" % (S(S_QP + 1), S(S_QP + 1)))
    for ds in range(8):
        e("global_load_dwordx4 %s, %s, %%[qptr] offset:%d" % (A(Q_["X"] + 4 * ds, 4), V(T3), ds * 32))
    for ds in range(8):
        e("global_load_dwordx4 %s, %s, %s offset:%d" % (A(Q_["Y"] + 4 * ds, 4), V(T3), S(S_QP, 2), ds * 32))
    e("v_mul_lo_u32 %s, %s, %%[dorowb]" % (V(T3), V(T0)))
    e("v_add_u32 %s, %s, %s" % (V(T3), V(T3), V(T2)))                 # dO: r pitch + 16 h
    e("s_lshl_b32 %s, %%[dorowb], 5" % S(S_X))
    e("s_mov_b64 %s, %%[doptr]" % S(S_DP2, 2))
    e("s_add_u32 %s, %s, %s" % (S(S_DP2), S(S_DP2), S(S_X)))
    e("s_addc_u32 %s, %s, 0" % (S(S_DP2 + 1), S(S_DP2 + 1)))
    for ds in range(8):
        e("global_load_dwordx4 %s, %s, %%[doptr] offset:%d" % (A(DO["X"] + 4 * ds, 4), V(T3), ds * 32))
    for ds in range(8):
        e("global_load_dwordx4 %s, %s, %s offset:%d" % (A(DO["Y"] + 4 * ds, 4), V(T3), S(S_DP2, 2), ds * 32))
    e("v_lshlrev_b32 %s, 2, %s" % (V(T2), V(T0)))                     # 4 r: the lane's entry of the statistics tables
    e("global_load_dword %s, %s, %%[lseptr]" % (V(NL["X"]), V(T2)))
    e("global_load_dword %s, %s, %%[lseptr] offset:128" % (V(NL["Y"]), V(T2)))
    e("global_load_dword %s, %s, %%[dltptr]" % (V(NDL["X"]), V(T2)))
    e("global_load_dword %s, %s, %%[dltptr] offset:128" % (V(NDL["Y"]), V(T2)))
    for r in range(128):
        e("v_accvgpr_write_b32 %s, 0" % A(r))
    e("s_waitcnt vmcnt(0)")
    for blk in "XY":
        for r in range(1, 16):
            e("v_mov_b32 %s, %s" % (V(NL[blk] + r), V(NL[blk])))
        for r in range(8):
            e("v_mov_b32 %s, 0" % V(DS[blk] + r))
    # the first phase's vector stream works on Y(-1, 1): scores of -1e30 (p = 0), dP = 0 -> dS_Y = 0
    for r in range(16):
        e("v_mov_b32 %s, 0xf149f2ca" % V(S_["Y"] + r))
        e("v_mov_b32 %s, 0" % V(DP["Y"] + r))
    for r in range(4):
        e("v_mov_b32 %s, 0" % V(HH4[r]))
    e("s_barrier")
    for i in range(8):
        e(kr_read(i, 0, 0))
        e(vr_read(i, 0, 0))
        L.extend(kt_reads(i, 0, 0))                                    # K^T(-1, 1) stands in: any finite operand (dS = 0)
    e("s_mov_b32 %s, 0" % S(S_T))
    if drop:
        L.extend(kh_reads(0))                                          # tile 0's; tile t + 1's are read in phase A2 of tile t
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 3")
    return L


def epilogue(drop):
    """dq = dQ^T * fs -> bf16, staged through LDS as whole rows (the ring is free behind the barrier), stored to dqkv's q part"""
    L = []
    e = L.append
    e("s_nop 15")
    e("s_nop 3")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")
    e("v_and_b32 %s, 31, %s" % (V(T0), V(VLANE)))
    e("v_lshrrev_b32 %s, 5, %s" % (V(T1), V(VLANE)))
    e("v_lshlrev_b32 %s, 8, %s" % (V(24), V(T0)))
    e("v_lshl_add_u32 %s, %s, 3, %s" % (V(24), V(T1), V(24)))
    e("s_lshl_b32 %s, %%[wave], 14" % S(S_X))
    e("s_add_u32 %s, %s, %s" % (S(S_X), S(S_X), S(S_LDS)))
    e("v_add_u32 %s, %s, %s" % (V(24), S(S_X), V(24)))
    e("v_and_b32 %s, 15, %s" % (V(T2), V(VLANE)))
    e("v_lshlrev_b32 %s, 4, %s" % (V(25), V(T2)))
    for c in range(16):
        e("v_xor_b32 %s, 0x%x, %s" % (V(32 + c), c << 4, V(25)))
        e("v_add_u32 %s, %s, %s" % (V(32 + c), V(32 + c), V(24)))
    for blk in "XY":
        for db in range(4):
            for g4 in range(4):
                base = DQ[blk] + 16 * db + 4 * g4
                for k in range(4):
                    e("v_accvgpr_read_b32 %s, %s" % (V(48 + k), A(base + k)))
                for k in range(4):
                    e("v_mul_f32 %s, %%[fs], %s" % (V(48 + k), V(48 + k)))
                e("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(52), V(48), V(49)))
                e("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(53), V(50), V(51)))
                e("ds_write_b64 %s, %s offset:%d" % (V(32 + 4 * db + g4), V(52, 2), (8192 if blk == "Y" else 0)))
    e("v_lshrrev_b32 %s, 4, %s" % (V(T0), V(VLANE)))
    e("v_and_b32 %s, 15, %s" % (V(T1), V(VLANE)))
    e("v_xor_b32 %s, %s, %s" % (V(T2), V(T1), V(T0)))
    e("v_lshlrev_b32 %s, 4, %s" % (V(T2), V(T2)))
    e("v_lshl_add_u32 %s, %s, 8, %s" % (V(T2), V(T0), V(T2)))
    e("v_add_u32 %s, %s, %s" % (V(T2), S(S_X), V(T2)))
    for k in range(4):
        e("v_xor_b32 %s, 0x%x, %s" % (V(28 + k), (4 * k) << 4, V(T2)))
    e("v_mul_lo_u32 %s, %s, %%[strideb]" % (V(T3), V(T0)))
    e("v_lshl_add_u32 %s, %s, 4, %s" % (V(T3), V(T1), V(T3)))
    e("s_waitcnt lgkmcnt(0)")
    e("s_mov_b64 %s, %%[optr]" % S(S_OP, 2))
    e("s_lshl_b32 %s, %%[strideb], 2" % S(S_Y2))
    for half in range(2):
        for i in range(8):
            ii = half * 8 + i
            e("ds_read_b128 %s, %s offset:%d" % (V(64 + 4 * i, 4), V(28 + (ii & 3)), 1024 * ii))
        e("s_waitcnt lgkmcnt(0)")
        for i in range(8):
            e("global_store_dwordx4 %s, %s, %s" % (V(T3), V(64 + 4 * i, 4), S(S_OP, 2)))
            e("s_add_u32 %s, %s, %s" % (S(S_OP), S(S_OP), S(S_Y2)))
            e("s_addc_u32 %s, %s, 0" % (S(S_OP + 1), S(S_OP + 1)))
    e("s_waitcnt vmcnt(0)")
    e("s_mov_b32 m0, %s" % S(S_M0))
    return L


BASE = dict(lag=1, dstride=3)


def gen(drop, cfg=None):
    cfg = BASE if cfg is None else cfg
    if cfg.get("abl_1616") and not cfg.get("_in1616"):      # timing only, see gen_attn_fwd.to1616
        return F.to1616(gen(drop, dict(cfg, _in1616=1)))
    L = prologue(drop)
    L.append("o2dq_loop_%=:")
    body = []
    for q in range(4):
        for ph in range(4):
            body += phase(ph, q, drop, cfg)
    # the re-fills of the last phase are outstanding when the body is entered again: the waits are placed for that state (the
    # first entry, from the prologue, has nothing outstanding: every wait is then trivially satisfied)
    _, carry = F.insert_lgkm_waits(body, ())
    body, pend = F.insert_lgkm_waits(body, carry)
    assert pend == carry, "the LDS reads outstanding at the end of the loop body differ from those at its entry"
    L += body
    L += ["s_cmp_lt_u32 %s, %%[nt]" % S(S_T), "s_cbranch_scc1 o2dq_loop_%="]
    # tail: dQ_X += K^T(nt-1, 1) dS_X(nt-1, 1) with Y(nt-1, 1)'s vector stream in its gaps, then the same product for Y
    # (slot of the last tile = 3: nt % 4 == 0; its key-group hashes are in the odd register set)
    tail = place(dq_only("X"), stream_gaps("Y", drop, KH[1] + 4, ("X", KH[0]))[:8], {})
    gl = stream_gaps("Y", drop, KH[1] + 4, ("X", KH[0]))
    tail += [x for g in gl[8:] for x in g]
    tail += ["s_nop 3"] + dq_only("Y")
    tail, pend = F.insert_lgkm_waits(tail, carry)
    assert not pend
    L += tail
    L += epilogue(drop)
    return L


def emit(path):
    out = ["// GENERATED by tools/gen_attn_dq.py -- do not edit; the schedule lives in that script.", "#pragma once",
           "#define O2_DQ_KH_OFF %d" % KH_OFF, "#define O2_DQ_LDS_BYTES(L) (%d + (L) + 64)" % KH_OFF]

    def macro(name, lines):
        out.append("#define %s \\" % name)
        for k, s in enumerate(lines):
            if s.endswith(":"):
                out.append('  "%s\\n"%s' % (s, " \\" if k + 1 < len(lines) else ""))
            else:
                out.append('  "%s\\n\\t"%s' % (s, " \\" if k + 1 < len(lines) else ""))

    macro("O2_DQ_ASM_DROP", gen(True))
    macro("O2_DQ_ASM_NODROP", gen(False))
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % r for r in range(256)] + ['"v%d"' % r for r in range(8, 256)] + \
           ['"s%d"' % r for r in range(36, 72)]
    out.append("#define O2_DQ_CLOBBERS \\")
    for k in range(0, len(clob), 16):
        chunk = ", ".join(clob[k:k + 16])
        out.append("  %s%s" % (chunk, ", \\" if k + 16 < len(clob) else ""))
    open(path, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    if "--cfg" in sys.argv:
        for kv in sys.argv[sys.argv.index("--cfg") + 1].split(","):
            k, v = kv.split("=")
            BASE[k] = int(v)
    if len(sys.argv) > 1 and sys.argv[1] == "show":
        lines = gen("nodrop" not in sys.argv)
        slot = -1
        for l in lines:
            if l.startswith("v_mfma"):
                slot += 1
            print(slot, l)
    else:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(root, "orbit-2_amd", "csrc", "attn_dq_asm.h")
        emit(out)
        print("wrote %s" % out, BASE)
