#!/usr/bin/env python3
"""Writes orbit-2_amd/csrc/attn_dkv_asm.h: the whole body of attn_bwd_dkv_w4_kernel (csrc/attn.hip) -- dK and dV of the attention
backward in one pass, d = 128, q stored pre-scaled, L % 256 == 0 -- as ONE `asm volatile` statement placed by this script (the
construction of tools/gen_attn_fwd.py / gen_attn_dq.py: read the first one's header for lane maps, LDS image and the emulator).

Math (autograd of components/attention.py:54-78; csrc/attn.hip attn_bwd_dkv128_kernel is the compiler-scheduled form), key on
the MFMA lane, query rows in the registers:
  S = Q~ K^T - lse2[q]   (accumulators START at the row statistics read from the tile's LDS table: p = exp2(S))
  dP' = dO V^T - delta[q] / dscale ;  dropped element: p_d = 0, dP' = -delta / dscale ;  dS = p * dP'
  dV^T += dO^T p_d ,  dK^T += Q^T dS ;   dk = dK * (ln 2 * dscale), dv = dV * dscale
Shape: workgroup = 4 waves = 128 keys of one (batch, head), one wave per SIMD; a wave owns 32 keys and sweeps all queries in
64-row tiles.  A UNIT = one 32-row half of a tile; phase(u) = 32 MFMAs:
      dV^T / dK^T += (unit u - 2) [16]   then   S / dP' chains of unit u [16]
with the vector stream of unit u - 1 (exp2, mask, products, packing) in its gaps 4..31.
  * registers: a[0:63] dK^T, a[64:127] dV^T, a[128:159] / a[160:191] the wave's K / V fragments (loaded once), a[192:223] /
    a[224:255] Q / dO ROW fragments of the unit whose S / dP' is being computed; v[32:63] / v[64:95] S and dP' of even / odd
    units, v[96:111] / v[112:127] their packed p_d and dS, v[128:159] / v[160:191] Q^T / dO^T fragments (transposing reads) of
    the unit whose dV / dK is being accumulated, v[192:207] the stream unit's -delta / dscale per element.
  * dropout: keep(row, key) = byte (key & 3) of mix(R(row) ^ K(key >> 2)) >= thr.  The four keys of a group are four adjacent
    lanes, so lane e' of a quad mixes row 8 g + 4 h + e' and every lane takes row e's word by `v_and_b32_dpp quad_perm:[e,e,e,e]`
    with its own byte mask: one mix per 4 elements.  R(row) of a tile's 64 rows is written to LDS by wave 1 two tiles ahead.
  * LDS: 4-slot ring, Q tiles at slot * 16 KiB, dO tiles 64 KiB up, one KiB of row statistics per slot behind them
    [-lse2 | -delta / dscale | R(row)]; tile t + 2 and its statistics are put in flight behind the one barrier of tile t.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_attn_fwd as F  # noqa: E402
from gen_attn_fwd import A, S, V  # noqa: E402

DK, DV, KF, VF, QROW, DOROW = 0, 64, 128, 160, 192, 224
S_ = [32, 64]
DP = [48, 80]
PD = [96, 112]
DS = [104, 120]
QT, DOT = 128, 160
NDL = [192, 8]          # -delta / dscale per element of the stream's unit: two sets (a unit's are fetched one phase ahead)
HM = 208
RHR = [212, 24]         # the quad's four row hashes, likewise
VRE, VRO, VT1, VT2, VRE2, VRO2, VT1B, VT2B = 216, 217, 218, 219, 220, 221, 222, 223
VDEQ, VDOQ, VDED, VDOD, VST, VSH, VBM, VTHS, VKEYH, VLANE, VL4 = 224, 225, 226, 227, 228, 229, 230, 231, 232, 233, 234
TD, T0, T1, T2, T3 = 235, 252, 253, 254, 255
S_DQ, S_DD, S_PCQ, S_PCD, S_LW, S_T, S_TBQ, S_TBD, S_NT1, S_OFQ, S_OFD, S_TMP, S_M0, S_MIX = 36, 40, 44, 48, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61
S_X, S_Y2, S_LDS, S_H2, S_OFS, S_RB, S_DL, S_DN, S_OP = 62, 63, 64, 65, 66, 67, 68, 72, 76

STAT_OFF = 131072
LDS_BYTES = STAT_OFF + 4096


def mfma(d, a, b, c):
    return "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (d, a, b, c)


def acc_mfmas(par):
    """dV^T / dK^T += of the unit whose packed p_d / dS sit in set `par` (fragment j = 4 ss + db in slots 2 j, 2 j + 1)"""
    out = []
    for j in range(8):
        ss, db = j >> 2, j & 3
        dv, dk = A(DV + 16 * db, 16), A(DK + 16 * db, 16)
        out.append(mfma(dv, V(DOT + 4 * j, 4), V(PD[par] + 4 * ss, 4), dv))
        out.append(mfma(dk, V(QT + 4 * j, 4), V(DS[par] + 4 * ss, 4), dk))
    return out


def sdp_mfmas(par):
    out = []
    s, dp = V(S_[par], 16), V(DP[par], 16)
    for ds in range(8):
        out.append(mfma(s, A(QROW + 4 * ds, 4), A(KF + 4 * ds, 4), s))
        out.append(mfma(dp, A(DOROW + 4 * ds, 4), A(VF + 4 * ds, 4), dp))
    return out


def row_reads(ds, slot, half):
    off = slot * 16384 + half * 8192 + 512 * (ds >> 1)
    return ["ds_read_b128 %s, %s offset:%d" % (A(QROW + 4 * ds, 4), V(VRO if ds & 1 else VRE), off),
            "ds_read_b128 %s, %s offset:%d" % (A(DOROW + 4 * ds, 4), V(VRO2 if ds & 1 else VRE2), off)]


def tr_reads(j, slot, half):
    ss, db = j >> 2, j & 3
    off = slot * 16384 + 2048 * (4 * half + 2 * ss) + 512 * db
    return ["ds_read_b64_tr_b16 %s, %s offset:%d" % (V(DOT + 4 * j, 2), V(VT1B), off),
            "ds_read_b64_tr_b16 %s, %s offset:%d" % (V(DOT + 4 * j + 2, 2), V(VT2B), off),
            "ds_read_b64_tr_b16 %s, %s offset:%d" % (V(QT + 4 * j, 2), V(VT1), off),
            "ds_read_b64_tr_b16 %s, %s offset:%d" % (V(QT + 4 * j + 2, 2), V(VT2), off)]


def init_reads(par, slot, half):
    """S / dP' accumulators of a unit start at its rows' statistics: registers 4 g .. 4 g + 3 <- table[32 half + 8 g + 4 h ..]"""
    out = []
    for g in range(4):
        off = slot * 1024 + (32 * half + 8 * g) * 4
        out.append("ds_read_b128 %s, %s offset:%d" % (V(S_[par] + 4 * g, 4), V(VST), off))
        out.append("ds_read_b128 %s, %s offset:%d" % (V(DP[par] + 4 * g, 4), V(VST), off + 256))
    return out


def stream_prep(par, slot, half, drop):
    """what the stream of the unit in set `par` needs from the statistics slot: -delta / dscale per element, the quad's four row
    hashes -- read ONE PHASE AHEAD of the stream, one instruction per gap (16 LDS reads bunched in two gaps stalled the issue)"""
    out = []
    if not drop:
        return out
    for g in range(4):
        off = slot * 1024 + (32 * half + 8 * g) * 4
        out.append("ds_read_b128 %s, %s offset:%d" % (V(NDL[par] + 4 * g, 4), V(VST), off + 256))
        out.append("ds_read_b32 %s, %s offset:%d" % (V(RHR[par] + g), V(VSH), off + 512))
    return out


def mix(g, par):
    return ["v_xor_b32 %s, %s, %s" % (V(HM + g), V(RHR[par] + g), V(VKEYH)),
            "v_mul_lo_u32 %s, %s, %s" % (V(HM + g), V(HM + g), S(S_MIX)),
            "v_xor_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" % (V(HM + g), V(HM + g), V(HM + g))]


def stream_gaps(par, drop, g0=4):
    """vector work of the unit in register set `par`, as 32 per-gap lists (gaps g0..31): element i in gap g0 + (27 i) // 16"""
    gaps = [[] for _ in range(32)]
    sb, db = S_[par], DP[par]
    eg = lambda i: g0 + ((31 - g0 - 1) * i) // 15
    if drop:
        m0 = mix(0, par)
        gaps[g0 - 2] += m0[:2]
        gaps[g0 - 1] += m0[2:]
    for i in range(16):
        g, gl, e = eg(i), i >> 2, i & 3
        if drop and e == 0 and gl + 1 < 4:
            for k, ins in enumerate(mix(gl + 1, par)):
                gaps[min(31, g + k)].append(ins)
        gaps[g].append("v_exp_f32 %s, %s" % (V(sb + i), V(sb + i)))
        if drop:
            gaps[g].append("v_and_b32_dpp %s, %s, %s quad_perm:[%d,%d,%d,%d] row_mask:0xf bank_mask:0xf" % (V(T3), V(HM + gl), V(VBM), e, e, e, e))
            gaps[g].append("v_cmp_ge_u32 vcc, %s, %s" % (V(T3), V(VTHS)))
            gaps[g].append("v_cndmask_b32 %s, %s, %s, vcc" % (V(db + i), V(NDL[par] + i), V(db + i)))
        gaps[g + 1].append("v_mul_f32 %s, %s, %s" % (V(db + i), V(sb + i), V(db + i)))
        if drop:
            gaps[g + 1].append("v_cndmask_b32 %s, 0, %s, vcc" % (V(sb + i), V(sb + i)))
        if i & 1:                                  # the pair's packed p_d and dS (operands of the unit's dV / dK MFMAs, next phase)
            k = i >> 1
            gaps[g + 1].append("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(PD[par] + k), V(sb + i - 1), V(sb + i)))
            gaps[g + 1].append("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(DS[par] + k), V(db + i - 1), V(db + i)))
    return gaps


def dma_piece(which, j, slot):
    sof, desc, base, pc, vde, vdo = (S_OFQ, S_DQ, 0, S_PCQ, VDEQ, VDOQ) if which == "Q" else (S_OFD, S_DD, 65536, S_PCD, VDED, VDOD)
    return ["s_add_u32 %s, %s, %s" % (S(S_TMP), S(sof), S(pc + j)),
            "s_add_u32 m0, %s, %d" % (S(S_LW), base + slot * 16384 + j * 1024),
            "s_nop 0",
            "buffer_load_dwordx4 %s, %s, %s offen lds" % (V(vdo if j >= 2 else vde), S(desc, 4), S(S_TMP))]


def tile_offsets(dt):
    return ["s_add_u32 %s, %s, %d" % (S(S_X), S(S_T), dt), "s_min_u32 %s, %s, %s" % (S(S_X), S(S_X), S(S_NT1)),
            "s_mul_i32 %s, %s, %s" % (S(S_OFQ), S(S_X), S(S_TBQ)), "s_mul_i32 %s, %s, %s" % (S(S_OFD), S(S_X), S(S_TBD)),
            "s_lshl_b32 %s, %s, 8" % (S(S_OFS), S(S_X)), "s_lshl_b32 %s, %s, 6" % (S(S_RB), S(S_X))]


def stats_stage(slot, label, drop):
    """wave 0: the tile's -lse2 / -delta rows by two dword LDS-DMA pieces; wave 1: R(row) of its 64 rows (o2_hash64 with the
    seed-dependent part folded into %[hseed] by the caller; row index = %[rowbase] + 64 t + lane)"""
    out = ["s_cmp_eq_u32 %[wave], 0", "s_cbranch_scc0 o2kv_s1_%s_%%=" % label,
           "s_add_u32 m0, %s, %d" % (S(S_LDS), STAT_OFF + slot * 1024), "s_nop 0",
           "buffer_load_dword %s, %s, %s offen lds" % (V(VL4), S(S_DL, 4), S(S_OFS)),
           "s_add_u32 m0, %s, %d" % (S(S_LDS), STAT_OFF + slot * 1024 + 256), "s_nop 0",
           "buffer_load_dword %s, %s, %s offen lds" % (V(VL4), S(S_DN, 4), S(S_OFS)),
           "o2kv_s1_%s_%%=:" % label]
    if drop:
        out += ["s_cmp_eq_u32 %[wave], 1", "s_cbranch_scc0 o2kv_s2_%s_%%=" % label,
                "s_add_u32 %s, %s, %%[rowbase]" % (S(S_X), S(S_RB)),
                "v_add_u32 %s, %s, %s" % (V(T0), S(S_X), V(VLANE)),
                "v_xor_b32 %s, %%[hseed], %s" % (V(T0), V(T0)),
                "v_lshrrev_b32 %s, 16, %s" % (V(T1), V(T0)), "v_xor_b32 %s, %s, %s" % (V(T0), V(T0), V(T1)),
                "v_mul_lo_u32 %s, %s, %s" % (V(T0), V(T0), S(S_Y2)),
                "v_lshrrev_b32 %s, 15, %s" % (V(T1), V(T0)), "v_xor_b32 %s, %s, %s" % (V(T0), V(T0), V(T1)),
                "v_mul_lo_u32 %s, %s, %s" % (V(T0), V(T0), S(S_H2)),
                "v_lshrrev_b32 %s, 16, %s" % (V(T1), V(T0)), "v_xor_b32 %s, %s, %s" % (V(T0), V(T0), V(T1)),
                "v_add_u32 %s, 0x%x, %s" % (V(T1), STAT_OFF + slot * 1024 + 512, V(VL4)),
                "v_add_u32 %s, %s, %s" % (V(T1), S(S_LDS), V(T1)),
                "ds_write_b32 %s, %s" % (V(T1), V(T0)),
                "o2kv_s2_%s_%%=:" % label]
    return out


def place(mf, gaps, fixed):
    out = []
    for m, ins in enumerate(mf):
        out.append(ins)
        out += fixed.get(m, [])
        if m < len(gaps):
            out += gaps[m]
    return out


def phase(u, drop, cfg):
    """phase of unit u (0..7 within the unrolled body of 4 tiles): slot = u >> 1, half = u & 1"""
    slot, half, par = (u >> 1) & 3, u & 1, u & 1
    mf = acc_mfmas(par) + sdp_mfmas(par)               # unit u - 2 has u's parity
    fixed = {}

    def add(g, ins):
        fixed.setdefault(g, []).extend(ins if isinstance(ins, list) else [ins])

    # stream: unit u - 1 (other parity); its tile slot / half
    su = (u - 1) % 8
    s_slot, s_half = (su >> 1) & 3, su & 1
    gaps = stream_gaps(par ^ 1, drop)
    if cfg.get("abl_valu"):
        gaps = [[] for _ in range(32)]
    for k, ins in enumerate(init_reads(par, slot, half)):            # needed by the chains' first MFMAs (slot 16)
        add(k, ins)
    for k, ins in enumerate(stream_prep(par, slot, half, drop)):       # for unit u's stream, which runs in the NEXT phase
        add(18 + k, ins)
    if half == 0:
        # tile t's barrier: tile t + 1 (and its statistics) have landed for every wave; the slot written next held tile t - 2
        add(2, ["s_waitcnt vmcnt(0)", "s_barrier"] + tile_offsets(2))
        g = 4
        for w_ in ("Q", "D"):
            for j in range(4):
                add(g, dma_piece(w_, j, (slot + 2) & 3))
                g += cfg["dstride"]
        add(g, stats_stage((slot + 2) & 3, "u%d" % u, drop))
    # re-fills: Q^T / dO^T of unit u - 1 behind the dV / dK MFMAs (slots 0..15), Q / dO rows of unit u + 1 behind the chains
    for j in range(8):
        tr = tr_reads(j, s_slot, s_half)
        add(2 * j + 1 + cfg["lag"], tr[:2])
        add(2 * j + 2 + cfg["lag"], tr[2:])
    nu = (u + 1) % 8
    for ds in range(8):
        add(min(31, 16 + 2 * ds + 1 + cfg["lag"]), row_reads(ds, (nu >> 1) & 3, nu & 1))
    if half == 1:
        add(31, "s_add_u32 %s, %s, 1" % (S(S_T), S(S_T)))
    return place(mf, gaps, fixed)


def lane_addresses(e, L):
    """prologue part shared in spirit with gen_attn_fwd.prologue: lane-constant LDS and DMA addresses"""
    e("v_mbcnt_lo_u32_b32 %s, -1, 0" % V(VLANE))
    e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (V(VLANE), V(VLANE)))
    e("s_mov_b32 %s, %%[ldsb]" % S(S_LDS))
    e("v_and_b32 %s, 31, %s" % (V(T0), V(VLANE)))
    e("v_lshrrev_b32 %s, 5, %s" % (V(T1), V(VLANE)))
    e("v_lshrrev_b32 %s, 3, %s" % (V(T2), V(T0)))
    e("v_lshlrev_b32 %s, 11, %s" % (V(VRE), V(T2)))
    e("v_and_b32 %s, 7, %s" % (V(T2), V(T0)))
    e("v_lshl_add_u32 %s, %s, 6, %s" % (V(VRE), V(T2), V(VRE)))
    e("v_bfe_u32 %s, %s, 2, 2" % (V(T2), V(T0)))
    e("v_xor_b32 %s, %s, %s" % (V(T2), V(T2), V(T1)))
    e("v_lshl_add_u32 %s, %s, 4, %s" % (V(VRE), V(T2), V(VRE)))
    e("v_add_u32 %s, %s, %s" % (V(VRE), S(S_LDS), V(VRE)))
    e("v_xor_b32 %s, 32, %s" % (V(VRO), V(VRE)))
    e("v_add_u32 %s, 0x10000, %s" % (V(VRE2), V(VRE)))
    e("v_add_u32 %s, 0x10000, %s" % (V(VRO2), V(VRO)))
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
    e("v_add_u32 %s, 0x10000, %s" % (V(VT1B), V(VT1)))
    e("v_add_u32 %s, 0x10000, %s" % (V(VT2B), V(VT2)))
    # DMA source offsets: ((lane >> 2) & 7) * pitch + 16 (4 (lane >> 5) + ((lane & 3) ^ ((lane >> 4) & 1))), odd pieces ^ 32
    for pitch, vde, vdo in (("%[strideb]", VDEQ, VDOQ), ("%[dorowb]", VDED, VDOD)):
        e("v_bfe_u32 %s, %s, 2, 3" % (V(T2), V(VLANE)))
        e("v_mul_lo_u32 %s, %s, %s" % (V(vde), V(T2), pitch))
        e("v_bfe_u32 %s, %s, 4, 1" % (V(T2), V(VLANE)))
        e("v_and_b32 %s, 3, %s" % (V(T3), V(VLANE)))
        e("v_xor_b32 %s, %s, %s" % (V(T2), V(T2), V(T3)))
        e("v_lshl_add_u32 %s, %s, 2, %s" % (V(T2), V(T1), V(T2)))
        e("v_lshl_add_u32 %s, %s, 4, %s" % (V(vde), V(T2), V(vde)))
        e("v_xor_b32 %s, 32, %s" % (V(vdo), V(vde)))
    # statistics: 16-byte rows at + 16 h; the quad's row hash at + 4 (4 h + (lane & 3))
    e("v_lshlrev_b32 %s, 4, %s" % (V(VST), V(T1)))
    e("v_add_u32 %s, 0x%x, %s" % (V(VST), STAT_OFF, V(VST)))
    e("v_add_u32 %s, %s, %s" % (V(VST), S(S_LDS), V(VST)))
    e("v_and_b32 %s, 3, %s" % (V(T2), V(VLANE)))
    e("v_lshl_add_u32 %s, %s, 2, %s" % (V(T3), V(T1), V(T2)))        # 4 h + (lane & 3)
    e("v_lshlrev_b32 %s, 2, %s" % (V(VSH), V(T3)))
    e("v_add_u32 %s, 0x%x, %s" % (V(VSH), STAT_OFF, V(VSH)))
    e("v_add_u32 %s, %s, %s" % (V(VSH), S(S_LDS), V(VSH)))
    e("v_lshlrev_b32 %s, 3, %s" % (V(T2), V(T2)))                     # 8 (lane & 3): this lane's byte of a mask word
    e("v_mov_b32 %s, 0xff" % V(T3))
    e("v_lshlrev_b32 %s, %s, %s" % (V(VBM), V(T2), V(T3)))
    e("v_mov_b32 %s, %%[thr]" % V(T3))
    e("v_lshlrev_b32 %s, %s, %s" % (V(VTHS), V(T2), V(T3)))
    e("v_mov_b32 %s, %%[keyh]" % V(VKEYH))
    e("v_lshlrev_b32 %s, 2, %s" % (V(VL4), V(VLANE)))


def prologue(drop):
    L = []
    e = L.append
    e("s_nop 4")
    e("s_mov_b32 %s, m0" % S(S_M0))
    lane_addresses(e, L)
    e("s_mov_b32 %s, 0x9E3779B1" % S(S_MIX))
    e("s_mov_b32 %s, 0x7FEB352D" % S(S_Y2))
    e("s_mov_b32 %s, 0x846CA68B" % S(S_H2))
    for desc, ptr in ((S_DQ, "%[qptr]"), (S_DD, "%[doptr]"), (S_DL, "%[lseptr]"), (S_DN, "%[dltptr]")):
        e("s_mov_b64 %s, %s" % (S(desc, 2), ptr))
        e("s_mov_b32 %s, 0x7fffffff" % S(desc + 2))
        e("s_mov_b32 %s, 0x00020000" % S(desc + 3))
    e("s_lshl_b32 %s, %%[strideb], 6" % S(S_TBQ))
    e("s_lshl_b32 %s, %%[dorowb], 6" % S(S_TBD))
    e("s_sub_u32 %s, %%[nt], 1" % S(S_NT1))
    for pc, pitch in ((S_PCQ, "%[strideb]"), (S_PCD, "%[dorowb]")):
        e("s_lshl_b32 %s, %%[wave], 4" % S(S_X))
        e("s_mul_i32 %s, %s, %s" % (S(pc), S(S_X), pitch))
        e("s_add_u32 %s, %s, 128" % (S(pc + 1), S(pc)))
        e("s_lshl_b32 %s, %s, 3" % (S(S_X), pitch))
        e("s_add_u32 %s, %s, %s" % (S(pc + 2), S(pc), S(S_X)))
        e("s_add_u32 %s, %s, 128" % (S(pc + 3), S(pc + 2)))
    e("s_lshl_b32 %s, %%[wave], 12" % S(S_X))
    e("s_add_u32 %s, %s, %s" % (S(S_LW), S(S_LDS), S(S_X)))
    # tiles 0, 1 -> slots 0, 1 (+ tile 0 once more into slot 3: the stand-in for "tile -1" of the first re-fills)
    for tile, slot in ((0, 0), (1, 1), (0, 3)):
        e("s_mov_b32 %s, %d" % (S(S_T), tile))
        L.extend(tile_offsets(0))
        for w_ in ("Q", "D"):
            for j in range(4):
                L.extend(dma_piece(w_, j, slot))
        L.extend(stats_stage(slot, "p%d" % slot, drop))
    # the wave's K / V rows as MFMA B operands (key on the lane): lane (key r, h) holds elements 16 ds + 8 h .. + 7
    e("v_and_b32 %s, 31, %s" % (V(T0), V(VLANE)))
    e("v_lshrrev_b32 %s, 5, %s" % (V(T1), V(VLANE)))
    e("v_lshlrev_b32 %s, 4, %s" % (V(T2), V(T1)))
    e("v_mul_lo_u32 %s, %s, %%[strideb]" % (V(T3), V(T0)))
    e("v_add_u32 %s, %s, %s" % (V(T3), V(T3), V(T2)))
    for ds in range(8):
        e("global_load_dwordx4 %s, %s, %%[kptr] offset:%d" % (A(KF + 4 * ds, 4), V(T3), ds * 32))
    for ds in range(8):
        e("global_load_dwordx4 %s, %s, %%[vptr] offset:%d" % (A(VF + 4 * ds, 4), V(T3), ds * 32))
    for r in range(128):
        e("v_accvgpr_write_b32 %s, 0" % A(r))
    for par in range(2):
        for r in range(8):
            e("v_mov_b32 %s, 0" % V(PD[par] + r))
            e("v_mov_b32 %s, 0" % V(DS[par] + r))
    # phase 0's stream works on "unit -1" in set 1: scores of -1e30 (p = 0), dP' = 0 -> p_d = dS = 0
    for r in range(16):
        e("v_mov_b32 %s, 0xf149f2ca" % V(S_[1] + r))
        e("v_mov_b32 %s, 0" % V(DP[1] + r))
        e("v_mov_b32 %s, 0" % V(NDL[1] + r))
    for r in range(4):
        e("v_mov_b32 %s, 0" % V(HM + r))
        e("v_mov_b32 %s, 0" % V(RHR[1] + r))
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")
    for ds in range(8):
        L.extend(row_reads(ds, 0, 0))                                 # unit 0's rows
    for j in range(8):
        L.extend(tr_reads(j, 0, 0))                                   # stand-in for unit -2 (its p_d / dS are 0)
    e("s_mov_b32 %s, 0" % S(S_T))
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 3")
    return L


def epilogue():
    """dk = dK^T * fk, dv = dV^T * fv -> bf16, staged through LDS as whole rows, stored to the k / v thirds of dqkv"""
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
    for base, scale, off in ((DK, "%[fk]", 0), (DV, "%[fv]", 8192)):
        for db in range(4):
            for g4 in range(4):
                r0 = base + 16 * db + 4 * g4
                for k in range(4):
                    e("v_accvgpr_read_b32 %s, %s" % (V(48 + k), A(r0 + k)))
                for k in range(4):
                    e("v_mul_f32 %s, %s, %s" % (V(48 + k), scale, V(48 + k)))
                e("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(52), V(48), V(49)))
                e("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(53), V(50), V(51)))
                e("ds_write_b64 %s, %s offset:%d" % (V(32 + 4 * db + g4), V(52, 2), off))
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
    e("s_lshl_b32 %s, %%[strideb], 2" % S(S_Y2))
    for ptr, off in (("%[okptr]", 0), ("%[ovptr]", 8192)):
        e("s_mov_b64 %s, %s" % (S(S_OP, 2), ptr))
        for i in range(8):
            e("ds_read_b128 %s, %s offset:%d" % (V(64 + 4 * i, 4), V(28 + (i & 3)), off + 1024 * i))
        e("s_waitcnt lgkmcnt(0)")
        for i in range(8):
            e("global_store_dwordx4 %s, %s, %s" % (V(T3), V(64 + 4 * i, 4), S(S_OP, 2)))
            e("s_add_u32 %s, %s, %s" % (S(S_OP), S(S_OP), S(S_Y2)))
            e("s_addc_u32 %s, %s, 0" % (S(S_OP + 1), S(S_OP + 1)))
    e("s_waitcnt vmcnt(0)")
    e("s_mov_b32 m0, %s" % S(S_M0))
    return L


BASE = dict(lag=0, dstride=2)


def gen(drop, cfg=None):
    cfg = BASE if cfg is None else cfg
    L = prologue(drop)
    L.append("o2kv_loop_%=:")
    body = []
    for u in range(8):
        body += phase(u, drop, cfg)
    _, carry = F.insert_lgkm_waits(body, ())
    body, pend = F.insert_lgkm_waits(body, carry)
    assert pend == carry, "the LDS reads outstanding at the end of the loop body differ from those at its entry"
    L += body
    L += ["s_cmp_lt_u32 %s, %%[nt]" % S(S_T), "s_cbranch_scc1 o2kv_loop_%="]
    # tail (units 2 nt - 2 in set 0 ... wait for its packs; 2 nt - 1 in set 1): phase 2 nt = dV / dK of unit 2 nt - 2 with the
    # stream of unit 2 nt - 1 in its gaps and that unit's transposed fragments re-filled; then dV / dK of unit 2 nt - 1
    fixed = {}
    for j in range(8):
        tr = tr_reads(j, 3, 1)
        fixed.setdefault(2 * j + 1, []).extend(tr[:2])
        fixed.setdefault(min(15, 2 * j + 2), []).extend(tr[2:])
    g = stream_gaps(1, drop)
    t0 = place(acc_mfmas(0), g[:16], fixed)
    t0 += [x for gg in g[16:] for x in gg + ["s_nop 0"]] + ["s_nop 3"]      # (no MFMA between these gaps: keep exp2 results one slot apart)
    t0 += acc_mfmas(1)
    t0, pend = F.insert_lgkm_waits(t0, carry)
    assert not pend
    L += t0
    L += epilogue()
    return L


def emit(path):
    out = ["// GENERATED by tools/gen_attn_dkv.py -- do not edit; the schedule lives in that script.", "#pragma once",
           "#define O2_KV_STAT_OFF %d" % STAT_OFF, "#define O2_KV_LDS_BYTES %d" % LDS_BYTES]

    def macro(name, lines):
        out.append("#define %s \\" % name)
        for k, s in enumerate(lines):
            if s.endswith(":"):
                out.append('  "%s\\n"%s' % (s, " \\" if k + 1 < len(lines) else ""))
            else:
                out.append('  "%s\\n\\t"%s' % (s, " \\" if k + 1 < len(lines) else ""))

    macro("O2_KV_ASM_DROP", gen(True))
    macro("O2_KV_ASM_NODROP", gen(False))
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % r for r in range(256)] + ['"v%d"' % r for r in range(8, 256)] + \
           ['"s%d"' % r for r in range(36, 78)]
    out.append("#define O2_KV_CLOBBERS \\")
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
        out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(root, "orbit-2_amd", "csrc", "attn_dkv_asm.h")
        emit(out)
        print("wrote %s" % out, BASE)
