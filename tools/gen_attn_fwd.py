#!/usr/bin/env python3
"""Writes orbit-2_amd/csrc/attn_fwd_asm.h: the whole body of attn_fwd_w4_kernel (csrc/attn.hip) as ONE `asm volatile` statement whose
instruction stream -- every MFMA, LDS read, LDS-DMA piece, vector instruction, counted wait and barrier -- is placed by this script.
`tools/cdna_emu.py` executes the same text on the CPU (tests/test_attn_asm_emu_cpu.py).

Reference: components/attention.py:54-78 (scaled-dot-product attention with dropout on P), d = 128, q stored pre-scaled by
log2(e)/sqrt(d) (ORBIT2_ATTN_Q_PRESCALED), L % 256 == 0.

Shape (gfx950, one wave per SIMD, 512 registers per lane):
  * workgroup = 4 waves = 256 query rows of one (batch, head); a wave = 64 rows = two 32-row blocks X, Y.
  * all products are v_mfma_f32_32x32x16_bf16 with the query on the lane: S^T = K Q^T (A = K rows from LDS, B = Q fragments kept
    in accumulator registers), O^T += V^T P^T (A = transposed V reads, B = P^T packed from the S registers, k order
    16 s + 8 (j >> 2) + 4 h + (j & 3)).
  * registers: a[0:63] / a[64:127] O^T of X / Y; a[128:159] / a[160:191] Q fragments of X / Y; a[192:255] the 16 K fragments of
    one 64-key tile; v[32:63] / v[64:95] S^T of X / Y; v[96:111] / v[112:127] -reference of X / Y (the C operand of each S
    chain's first MFMA: scores leave the matrix pipe as exp2 arguments); v[128:143] / v[144:159] P^T of X / Y (bf16 pairs);
    v[160:223] the 16 V^T fragments of one tile; the rest scratch.
  * per 64-key tile t two PHASES of 32 MFMAs:
      phase 1: S_Y(t) [16] + O_Y += V(t-1)^T P_Y(t-1)^T [16]   with the softmax / dropout of X's tile t in the gaps
      phase 2: S_X(t+1) [16] + O_X += V(t)^T P_X(t)^T [16]     with the softmax / dropout of Y's tile t in the gaps
    X runs half a tile ahead of Y, so every K / V fragment read from LDS feeds two MFMAs (K(t+1): X in phase 2, Y in the next
    phase 1; V(t): X in phase 2, Y in the next phase 1) and is re-filled in place right after its second use.
  * softmax with a LAZY reference (attn_fwd_lazy_kernel's formulation): the reference of a row is its maximum over the first
    tile and stays fixed; p = exp2(s) with no subtraction, maximum or rescale in the loop.  The tile's row sums are the guard:
    if one leaves [0, 2^40] (inf / NaN included) the wave branches to a fix-up that moves the block's reference (O, l, the
    -reference block and the tile's S), redoes the tile's probabilities and returns.
  * dropout: keep(row, key) = byte (key & 3) of mix(R(row) ^ K(key >> 2)) >= thr (csrc/common.h), R per lane, K from an LDS
    table of the whole sequence written by the kernel's prologue; l sums the un-dropped p, O the dropped ones, 1/(1-p) folded
    into the final scale.
  * K / V tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds) into a 2-slot ring of [K 16 KiB | V 16 KiB]; LDS image of a tile
    = 8-row x 32-column subtiles of 512 B (off = 2048 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3))):
    every row read and every transposed read is one of 2 + 2 lane-constant address registers plus an immediate.  One barrier
    per tile (phase 2): behind it K(t+3) and V(t+2) are put in flight one piece every few MFMAs; they are waited for at the
    next barrier, one tile before their first read.
"""
import os
import sys

# ---- registers -------------------------------------------------------------------------------------------------------
O_ = {"X": 0, "Y": 64}
Q_ = {"X": 128, "Y": 160}
AK = 192
S_ = {"X": 32, "Y": 64}
NR = {"X": 96, "Y": 112}
P_ = {"X": 128, "Y": 144}
VV = 160
KH = [224, 8]     # 8 key-group hashes of the tile: even tiles in v[224:231], odd tiles in v[8:15]
HT = 234          # shift temporary
PT = [236, 240]   # two sets of 4 exp results
ACA, ACB, PSUM = 244, 245, 248
L_ = {"X": 246, "Y": 247}
RH = {"X": 249, "Y": 250}   # row hashes (copied from the inputs)
VTHR = 251
T0, T1, T2, T3 = 252, 253, 254, 255
VKE, VKO, VV1, VV2, VDE, VDO, VKH, VLANE = 24, 25, 26, 27, 28, 29, 30, 31
# scalars (all clobbered)
S_DK, S_DV, S_PC, S_LW, S_T, S_TB, S_NT1, S_OFK, S_OFV, S_TMP, S_M0, S_MIX, S_RET = 36, 40, 44, 48, 49, 50, 51, 52, 53, 54, 56, 57, 58
S_QP, S_OP, S_X, S_Y2, S_EX, S_LDS = 60, 62, 64, 65, 66, 68

KH_OFF = 65536            # byte offset of the key-group hash table behind the 2 x 32 KiB ring
LIMIT = "0x53800000"      # 2^40


def V(b, n=1):
    return "v%d" % b if n == 1 else "v[%d:%d]" % (b, b + n - 1)


def A(b, n=1):
    return "a%d" % b if n == 1 else "a[%d:%d]" % (b, b + n - 1)


def S(b, n=1):
    return "s%d" % b if n == 1 else "s[%d:%d]" % (b, b + n - 1)


COST = {"v_exp_f32": 8, "v_mul_lo_u32": 8, "v_log_f32": 8, "v_rcp_f32": 8}


def cost(text):
    op = text.split()[0]
    if op.startswith("ds_") or op.startswith("s_"):
        return 1
    return COST.get(op, 4)


# ---- the vector stream of one block's tile: S -> p, row sums, dropout, P^T fragments --------------------------------------
HH3 = [232, 233, 235]     # mix values of three key groups in flight


def softmax_gaps(blk, drop, kh):
    """the stream as 32 per-gap lists, ONE v_exp_f32 per MFMA gap (two 8-cycle instructions in one gap do not hide behind the
    MFMA, MI355X_MICROARCH 'single-issue instructions HIDDEN per gap'): gap i holds exp of element i, the row-sum add of
    element i - 1, the mask compare + select of element i - 2, the bf16 pack of a finished pair and one instruction of the
    mix of key group i / 4 + 1.  S registers are only READ (the fix-up needs them intact); kh = first register of the tile's 8
    key-group hashes.  Element i = S register (i >> 4) * 16 + (i & 15): key group g = i >> 2."""
    sb, pb = S_[blk], P_[blk]
    gaps = [[] for _ in range(32)]
    pt = lambda i: 236 + (i & 7)

    def hashg(g):
        hh = HH3[g % 3]
        return ["v_xor_b32 %s, %s, %s" % (V(hh), V(RH[blk]), V(kh + g)),
                "v_mul_lo_u32 %s, %s, %s" % (V(hh), V(hh), S(S_MIX)),
                "v_xor_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" % (V(hh), V(hh), V(hh))]     # x ^= x >> 16 in one instruction

    def tail_ops(i):
        """what follows element i's exp two gaps later: mask, and the pack of its pair when i is odd"""
        o = []
        if drop:
            o.append("v_cmp_ge_u32_sdwa vcc, %s, %s src0_sel:BYTE_%d src1_sel:DWORD" % (V(HH3[(i >> 2) % 3]), V(VTHR), i & 3))
            o.append("v_cndmask_b32 %s, 0, %s, vcc" % (V(pt(i)), V(pt(i))))
        if i & 1:
            o.append("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(pb + (i >> 1)), V(pt(i - 1)), V(pt(i))))
        return o

    def add_op(i):
        # (v_pk_add_f32 on the exp-result pairs was measured: 16 instead of 31 instructions, 3.6-4.8 % SLOWER)
        if i == 1:
            return ["v_add_f32 %s, %s, %s" % (V(ACA), V(pt(0)), V(pt(1)))]
        if i == 3:
            return ["v_add_f32 %s, %s, %s" % (V(ACB), V(pt(2)), V(pt(3)))]
        if i in (0, 2):
            return []
        acc = ACA if i % 2 == 0 else ACB
        return ["v_add_f32 %s, %s, %s" % (V(acc), V(acc), V(pt(i)))]

    if drop:
        h0 = hashg(0)
        gaps[0] += h0[:2]
        gaps[1] += h0[2:]
    for i in range(32):
        if drop and i // 4 + 1 < 8 and i % 4 < 3:
            gaps[i].append(hashg(i // 4 + 1)[i % 4])
        gaps[i].append("v_exp_f32 %s, %s" % (V(pt(i)), V(sb + i)))
        if i >= 1:
            gaps[i] += add_op(i - 1)
        if i >= 2:
            gaps[i] += tail_ops(i - 2)
    gaps[31] += ["v_add_f32 %s, %s, %s" % (V(ACB), V(ACB), V(pt(31)))] + tail_ops(30) + tail_ops(31)
    gaps[31].append("v_add_f32 %s, %s, %s" % (V(PSUM), V(ACA), V(ACB)))
    return gaps


def softmax_stream(blk, drop, kh):
    return [x for g in softmax_gaps(blk, drop, kh) for x in g]


def guard(blk, site):
    return ["v_cmp_nge_f32 vcc, %s, %s" % (LIMIT, V(PSUM)),
            "s_cbranch_vccnz o2af_fix%d_%%=" % site,
            "o2af_ret%d_%%=:" % site,
            "v_add_f32 %s, %s, %s" % (V(L_[blk]), V(L_[blk]), V(PSUM))]


# ---- MFMA sequences ------------------------------------------------------------------------------------------------------
def mfma(d, a, b, c):
    return "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (d, a, b, c)


def s_chain(blk, cinit=None):
    """16 MFMAs S^T[kb] = K[kb] Q^T (+ C); K fragment f = 2 ds + kb in a[AK + 4 f]"""
    out = []
    for ds in range(8):
        for kb in range(2):
            d = V(S_[blk] + 16 * kb, 16)
            c = (cinit if cinit is not None else V(NR[blk], 16)) if ds == 0 else d
            out.append((mfma(d, A(AK + 4 * (2 * ds + kb), 4), A(Q_[blk] + 4 * ds, 4), c), ("K", 2 * ds + kb)))
    return out


def pv_chain(blk):
    """16 MFMAs O^T[db] += V^T[ks][db] P^T[ks]; V fragment j = 4 ks + db in v[VV + 4 j]"""
    out = []
    for ks in range(4):
        for db in range(4):
            d = A(O_[blk] + 16 * db, 16)
            out.append((mfma(d, V(VV + 4 * (4 * ks + db), 4), V(P_[blk] + 4 * ks, 4), d), ("V", 4 * ks + db)))
    return out


def k_read(f, slot):
    ds, kb = f >> 1, f & 1
    return "ds_read_b128 %s, %s offset:%d" % (A(AK + 4 * f, 4), V(VKO if ds & 1 else VKE), slot * 32768 + kb * 8192 + 512 * (ds >> 1))


def v_reads(j, slot):
    ks, db = j >> 2, j & 3
    kb, ss = ks >> 1, ks & 1
    off = slot * 32768 + 16384 + 2048 * (4 * kb + 2 * ss) + 512 * db
    return ["ds_read_b64_tr_b16 %s, %s offset:%d" % (V(VV + 4 * j, 2), V(VV1), off),
            "ds_read_b64_tr_b16 %s, %s offset:%d" % (V(VV + 4 * j + 2, 2), V(VV2), off)]


def kh_reads(par):
    return ["ds_read_b128 %s, %s" % (V(KH[par], 4), V(VKH)), "ds_read_b128 %s, %s offset:16" % (V(KH[par] + 4, 4), V(VKH)),
            "v_add_u32 %s, 64, %s" % (V(VKH), V(VKH))]


def dma_piece(which, j, slot):
    """piece j (0..3) of this wave's share of a K ('K') or V ('V') tile into ring slot `slot`; soffset in S_TMP"""
    sof, desc, half = (S_OFK, S_DK, 0) if which == "K" else (S_OFV, S_DV, 16384)
    return ["s_add_u32 %s, %s, %s" % (S(S_TMP), S(sof), S(S_PC + j)),
            "s_add_u32 m0, %s, %d" % (S(S_LW), slot * 32768 + half + j * 1024),
            "s_nop 0",
            "buffer_load_dwordx4 %s, %s, %s offen lds" % (V(VDO if j >= 2 else VDE), S(desc, 4), S(S_TMP))]


def tile_offsets(dk, dv):
    """S_OFK / S_OFV = byte offset of tile min(t + dk, nt - 1) / min(t + dv, nt - 1)"""
    return ["s_add_u32 %s, %s, %d" % (S(S_X), S(S_T), dk), "s_min_u32 %s, %s, %s" % (S(S_X), S(S_X), S(S_NT1)),
            "s_mul_i32 %s, %s, %s" % (S(S_OFK), S(S_X), S(S_TB)),
            "s_add_u32 %s, %s, %d" % (S(S_X), S(S_T), dv), "s_min_u32 %s, %s, %s" % (S(S_X), S(S_X), S(S_NT1)),
            "s_mul_i32 %s, %s, %s" % (S(S_OFV), S(S_X), S(S_TB))]


# ---- one phase: 32 MFMAs with everything else placed in the gaps ----------------------------------------------------------
def place(mf, valu, fixed, cfg):
    """mf: 32 MFMA texts; valu: ordered vector stream; fixed: {gap: [instructions]} -> flat list.  The vector stream is spread
    over gaps g0..g1 in proportion to its issue cost, fixed instructions go first in their gap."""
    if valu and isinstance(valu[0], list):
        gaps = valu                              # already one list per gap
    else:
        g0, g1 = cfg.get("valu_first", 0), cfg.get("valu_last", 31)
        total = sum(cost(x) for x in valu if not x.endswith(":"))
        per = total / float(g1 - g0 + 1)
        gaps = [[] for _ in range(32)]
        acc = 0.0
        for x in valu:
            g = min(g1, g0 + int(acc / per)) if per > 0 else g0
            gaps[g].append(x)
            if not x.endswith(":"):
                acc += cost(x)
    out = []
    for m in range(32):
        out.append(mf[m])
        out += fixed.get(m, [])
        out += gaps[m]
    return out


def phase(which, par, drop, site, cfg, first=False):
    """which = 1: MFMAs of Y on tile t, vector stream of X; which = 2: MFMAs of X on tile t + 1 / t, vector stream of Y.
    par = t & 1 (slot of tile t)."""
    blk_m, blk_v = ("Y", "X") if which == 1 else ("X", "Y")
    sc, pv = s_chain(blk_m), pv_chain(blk_m)
    mf = [x for x, _ in sc] + [x for x, _ in pv]
    fixed = {}

    def add(g, ins):
        fixed.setdefault(g, []).extend(ins if isinstance(ins, list) else [ins])

    if cfg.get("spread"):
        valu = softmax_stream(blk_v, drop, KH[par]) + guard(blk_v, site)
    else:
        valu = softmax_gaps(blk_v, drop, KH[par])
        valu[31] = valu[31] + guard(blk_v, site)
    # timing-only ablations (results are wrong by construction; tools/mkvar_af.sh): what each part of the loop costs
    def keep(f):
        return [[x for x in g if f(x)] for g in valu] if valu and isinstance(valu[0], list) else [x for x in valu if f(x)]
    if cfg.get("abl_valu"):
        valu = keep(lambda x: x.endswith(":"))
    if cfg.get("abl_mask"):
        valu = keep(lambda x: not (x.startswith("v_cmp_ge_u32_sdwa") or x.startswith("v_cndmask")))
    if cfg.get("abl_hash"):
        valu = keep(lambda x: not (x.startswith("v_mul_lo") or x.startswith("v_xor") or x.startswith("v_lshrrev")))
    if cfg.get("abl_sacc"):                       # timing only: the S chain accumulates in accumulator registers (O's)
        mf = [x.replace("v[%d:%d]" % (S_[blk_m], S_[blk_m] + 15), "a[%d:%d]" % (O_[blk_m], O_[blk_m] + 15))
               .replace("v[%d:%d]" % (S_[blk_m] + 16, S_[blk_m] + 31), "a[%d:%d]" % (O_[blk_m] + 16, O_[blk_m] + 31))
               .replace("v[%d:%d]" % (NR[blk_m], NR[blk_m] + 15), "a[%d:%d]" % (O_[blk_m] + 32, O_[blk_m] + 47)) for x in mf]
    if which == 1 and cfg.get("abl_lds"):
        return place(mf, valu, fixed, cfg)
    if which == 1:
        for f in range(16):                      # K(t+1) into the fragment registers, each right behind its last use (slot f)
            add(f + cfg["klag"], k_read(f, par ^ 1))
        for j in range(16):                      # V(t) likewise (used in slot 16 + j)
            g = 16 + j + cfg["vlag"]
            if g <= 31:
                add(g, v_reads(j, par))
            else:
                add(31, v_reads(j, par))
    else:
        b = cfg["bar"]
        add(b, ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + ([] if cfg.get("abl_bar") else ["s_barrier"]) + tile_offsets(3, 2))
        if drop:                                  # key-group hashes of tile t + 1 into the other register set
            add(b + 1, kh_reads(par ^ 1))
        pieces = [("K", j, par ^ 1) for j in range(4)] + [("V", j, par) for j in range(4)]
        g = b + 2
        for k, (w, j, sl) in enumerate(pieces):
            if not cfg.get("abl_dma"):
                add(g, dma_piece(w, j, sl))
            g += cfg["dstride"]
        assert g - cfg["dstride"] <= 31, "pieces run past the phase"
        add(31, ["s_add_u32 %s, %s, 1" % (S(S_T), S(S_T))])
    return place(mf, valu, fixed, cfg)


# ---- counted LDS waits ---------------------------------------------------------------------------------------------------
def regs_of(tok):
    """registers named by an operand token -> set of ('v' | 'a' | 's', index)"""
    tok = tok.strip()
    out = set()
    if not tok or tok[0] not in "vas" or tok in ("vcc", "scc", "s_nop"):
        return out
    kind = tok[0]
    body = tok[1:]
    if body.startswith("["):
        lo, hi = body[1:-1].split(":")
        for r in range(int(lo), int(hi) + 1):
            out.add((kind, r))
    elif body.isdigit():
        out.add((kind, int(body)))
    return out


def operands(text):
    parts = text.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops = [x.strip() for x in parts[1].split(",")]
    last = ops[-1].split()
    if last:
        ops[-1] = last[0]
    return parts[0], ops


def insert_lgkm_waits(seq, carry=()):
    """seq: flat instruction list entered with the LDS reads `carry` (destination-register sets, oldest first) outstanding.  Inserts the minimal counted s_waitcnt lgkmcnt(n) in front of every instruction that
    touches a register an outstanding LDS read will still write (reads return in issue order; counts above 15 clamp)."""
    out, pend = [], [set(x) for x in carry]          # pend: list of destination-register sets, oldest first
    for ins in seq:
        op, ops = operands(ins)
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in ins:
                pend = []
            out.append(ins)
            continue
        if op.endswith(":"):
            out.append(ins)
            continue
        touched = set()
        for o in ops:
            touched |= regs_of(o)
        need = None
        for k, dst in enumerate(pend):
            if dst & touched:
                need = k
        if need is not None:
            n = len(pend) - 1 - need
            n = min(n, 15)
            out.append("s_waitcnt lgkmcnt(%d)" % n)
            pend = pend[len(pend) - n:] if n > 0 else []
        if op.startswith("ds_read"):
            pend.append(regs_of(ops[0]))
        elif op.startswith("ds_write"):
            pend.append(set())                      # an LDS store occupies a slot of the same counter
        out.append(ins)
    return out, pend


# ---- prologue / tail / epilogue / fix-ups ------------------------------------------------------------------------------------
def rowmax(dst, sb):
    o = ["v_max3_f32 %s, %s, %s, %s" % (V(dst), V(sb), V(sb + 1), V(sb + 2))]
    for k in range(14):
        o.append("v_max3_f32 %s, %s, %s, %s" % (V(dst), V(dst), V(sb + 3 + 2 * k), V(sb + 4 + 2 * k)))
    o.append("v_max_f32 %s, %s, %s" % (V(dst), V(dst), V(sb + 31)))
    return o


def both_halves(reg, tmp, op):
    """reg <- op(reg of this lane, reg of the lane 32 away), in every lane"""
    return ["v_mov_b32 %s, %s" % (V(tmp), V(reg)), "s_nop 1", "v_permlane32_swap_b32 %s, %s" % (V(reg), V(tmp)), "s_nop 1",
            "%s %s, %s, %s" % (op, V(reg), V(reg), V(tmp))]


def prologue(drop):
    L = []
    e = L.append
    e("s_nop 4")
    e("s_mov_b32 %s, m0" % S(S_M0))
    # ---- lane-constant addresses ----
    e("v_mbcnt_lo_u32_b32 %s, -1, 0" % V(VLANE))
    e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (V(VLANE), V(VLANE)))
    e("s_mov_b32 %s, %%[ldsb]" % S(S_LDS))
    # K row reads: r = lane & 31, h = lane >> 5: base = 2048 (r >> 3) + 64 (r & 7) + 16 (h ^ ((r >> 2) & 3)), odd k-steps ^ 32
    e("v_and_b32 %s, 31, %s" % (V(T0), V(VLANE)))                  # r
    e("v_lshrrev_b32 %s, 5, %s" % (V(T1), V(VLANE)))               # h
    e("v_lshrrev_b32 %s, 3, %s" % (V(T2), V(T0)))
    e("v_lshlrev_b32 %s, 11, %s" % (V(VKE), V(T2)))
    e("v_and_b32 %s, 7, %s" % (V(T2), V(T0)))
    e("v_lshl_add_u32 %s, %s, 6, %s" % (V(VKE), V(T2), V(VKE)))
    e("v_bfe_u32 %s, %s, 2, 2" % (V(T2), V(T0)))                   # (r >> 2) & 3
    e("v_xor_b32 %s, %s, %s" % (V(T2), V(T2), V(T1)))
    e("v_lshl_add_u32 %s, %s, 4, %s" % (V(VKE), V(T2), V(VKE)))
    e("v_add_u32 %s, %s, %s" % (V(VKE), S(S_LDS), V(VKE)))
    e("v_xor_b32 %s, 32, %s" % (V(VKO), V(VKE)))
    # V transposed reads: g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3:
    #   first block 64 (4 h + q) + 16 ((2 g1 + (p >> 1)) ^ h) + 8 (p & 1); second block (first ^ 32) + 2048
    e("v_bfe_u32 %s, %s, 2, 2" % (V(T2), V(VLANE)))                # q
    e("v_lshl_add_u32 %s, %s, 2, %s" % (V(T2), V(T1), V(T2)))      # 4 h + q
    e("v_lshlrev_b32 %s, 6, %s" % (V(VV1), V(T2)))
    e("v_bfe_u32 %s, %s, 4, 1" % (V(T2), V(VLANE)))                # g1
    e("v_bfe_u32 %s, %s, 1, 1" % (V(T3), V(VLANE)))                # p >> 1
    e("v_lshl_add_u32 %s, %s, 1, %s" % (V(T2), V(T2), V(T3)))      # 2 g1 + (p >> 1)
    e("v_xor_b32 %s, %s, %s" % (V(T2), V(T2), V(T1)))
    e("v_lshl_add_u32 %s, %s, 4, %s" % (V(VV1), V(T2), V(VV1)))
    e("v_and_b32 %s, 1, %s" % (V(T2), V(VLANE)))
    e("v_lshl_add_u32 %s, %s, 3, %s" % (V(VV1), V(T2), V(VV1)))
    e("v_add_u32 %s, %s, %s" % (V(VV1), S(S_LDS), V(VV1)))
    e("v_xor_b32 %s, 32, %s" % (V(VV2), V(VV1)))
    e("v_add_u32 %s, 0x800, %s" % (V(VV2), V(VV2)))
    # LDS-DMA source offsets of a piece: ((lane >> 2) & 7) stride + 16 (4 (lane >> 5) + ((lane & 3) ^ (2 f + ((lane >> 4) & 1)))), f = 0 / 1
    e("v_bfe_u32 %s, %s, 2, 3" % (V(T2), V(VLANE)))
    e("v_mul_lo_u32 %s, %s, %%[strideb]" % (V(VDE), V(T2)))
    e("v_bfe_u32 %s, %s, 4, 1" % (V(T2), V(VLANE)))
    e("v_and_b32 %s, 3, %s" % (V(T3), V(VLANE)))
    e("v_xor_b32 %s, %s, %s" % (V(T2), V(T2), V(T3)))
    e("v_lshl_add_u32 %s, %s, 2, %s" % (V(T2), V(T1), V(T2)))      # 4 h + x
    e("v_lshl_add_u32 %s, %s, 4, %s" % (V(VDE), V(T2), V(VDE)))
    e("v_xor_b32 %s, 32, %s" % (V(VDO), V(VDE)))
    # key-group hash table: half h reads its 8 values of tile t at KH_OFF + 64 t + 32 h
    e("v_lshlrev_b32 %s, 5, %s" % (V(VKH), V(T1)))
    e("v_add_u32 %s, %s, %s" % (V(VKH), S(S_LDS), V(VKH)))
    e("v_add_u32 %s, 0x%x, %s" % (V(VKH), KH_OFF, V(VKH)))
    e("v_mov_b32 %s, %%[rhx]" % V(RH["X"]))
    e("v_mov_b32 %s, %%[rhy]" % V(RH["Y"]))
    e("v_mov_b32 %s, %%[thr]" % V(VTHR))
    e("s_mov_b32 %s, 0x9E3779B1" % S(S_MIX))
    # ---- scalars: descriptors, piece offsets, tile size ----
    e("s_mov_b64 %s, %%[kptr]" % S(S_DK, 2))
    e("s_mov_b32 %s, 0x7fffffff" % S(S_DK + 2))
    e("s_mov_b32 %s, 0x00020000" % S(S_DK + 3))
    e("s_add_u32 %s, %s, %%[hd2]" % (S(S_DV), S(S_DK)))
    e("s_addc_u32 %s, %s, 0" % (S(S_DV + 1), S(S_DK + 1)))
    e("s_mov_b32 %s, 0x7fffffff" % S(S_DV + 2))
    e("s_mov_b32 %s, 0x00020000" % S(S_DV + 3))
    e("s_lshl_b32 %s, %%[strideb], 6" % S(S_TB))                    # bytes of a 64-token tile
    e("s_sub_u32 %s, %%[nt], 1" % S(S_NT1))
    # piece j of wave w: tile piece i = 4 w + j: rows 8 (i >> 1) .., column half i & 1
    e("s_lshl_b32 %s, %%[wave], 4" % S(S_X))                        # 16 w = 8 * (2 w)
    e("s_mul_i32 %s, %s, %%[strideb]" % (S(S_PC), S(S_X)))
    e("s_add_u32 %s, %s, 128" % (S(S_PC + 1), S(S_PC)))
    e("s_lshl_b32 %s, %%[strideb], 3" % S(S_X))
    e("s_add_u32 %s, %s, %s" % (S(S_PC + 2), S(S_PC), S(S_X)))
    e("s_add_u32 %s, %s, 128" % (S(S_PC + 3), S(S_PC + 2)))
    e("s_lshl_b32 %s, %%[wave], 12" % S(S_X))
    e("s_add_u32 %s, %s, %s" % (S(S_LW), S(S_LDS), S(S_X)))
    # ---- tiles 0 and 1 -> slots 0 and 1; Q fragments -> accumulator registers ----
    for tile in range(2):
        e("s_mov_b32 %s, %d" % (S(S_T), tile))
        L.extend(["s_min_u32 %s, %s, %s" % (S(S_X), S(S_T), S(S_NT1)), "s_mul_i32 %s, %s, %s" % (S(S_OFK), S(S_X), S(S_TB)),
                  "s_mov_b32 %s, %s" % (S(S_OFV), S(S_OFK))])
        for w in ("K", "V"):
            for j in range(4):
                L.extend(dma_piece(w, j, tile))
    e("v_lshlrev_b32 %s, 4, %s" % (V(T2), V(T1)))                   # 16 h
    e("v_mul_lo_u32 %s, %s, %%[strideb]" % (V(T3), V(T0)))          # r stride
    e("v_add_u32 %s, %s, %s" % (V(T3), V(T3), V(T2)))
    e("s_lshl_b32 %s, %%[strideb], 5" % S(S_X))
    e("s_mov_b64 %s, %%[qptr]" % S(S_QP, 2))
    e("s_add_u32 %s, %s, %s" % (S(S_QP), S(S_QP), S(S_X)))
    e("s_addc_u32 %s, %s, 0" % (S(S_QP + 1), S(S_QP + 1)))
    for ds in range(8):
        e("global_load_dwordx4 %s, %s, %%[qptr] offset:%d" % (A(Q_["X"] + 4 * ds, 4), V(T3), ds * 32))
    for ds in range(8):
        e("global_load_dwordx4 %s, %s, %s offset:%d" % (A(Q_["Y"] + 4 * ds, 4), V(T3), S(S_QP, 2), ds * 32))
    for r in range(128):
        e("v_accvgpr_write_b32 %s, 0" % A(r))
    for blk in "XY":
        e("v_mov_b32 %s, 0" % V(L_[blk]))
        for r in range(16):
            e("v_mov_b32 %s, 0" % V(P_[blk] + r))
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    # ---- K(0), V(0) fragments; reference = row maximum over tile 0 ----
    for f in range(16):
        e(k_read(f, 0))
    for j in range(16):
        L.extend(v_reads(j, 0))
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")                                                  # every wave has read slot 0's K: tile 2 may land there
    L.extend(["s_mov_b32 %s, 2" % S(S_T), "s_min_u32 %s, %s, %s" % (S(S_X), S(S_T), S(S_NT1)),
              "s_mul_i32 %s, %s, %s" % (S(S_OFK), S(S_X), S(S_TB))])
    for j in range(4):
        L.extend(dma_piece("K", j, 0))
    for blk in "XY":
        L.extend(x for x, _ in s_chain(blk, cinit="0"))
    e("s_nop 15")
    e("s_nop 3")
    for blk in "XY":
        L.extend(rowmax(T0, S_[blk]))
        L.extend(both_halves(T0, T1, "v_max_f32"))
        e("v_mul_f32 %s, -1.0, %s" % (V(NR[blk]), V(T0)))
        for r in range(1, 16):
            e("v_mov_b32 %s, %s" % (V(NR[blk] + r), V(NR[blk])))
    e("s_nop 3")
    L.extend(x for x, _ in s_chain("X"))                             # S_X(0) relative to the reference: the loop's entry state
    e("s_nop 15")                                                    # (phase 1's first vector instruction reads it)
    e("s_nop 3")
    e("s_mov_b32 %s, 0" % S(S_T))
    if drop:
        L.extend(kh_reads(0))                                        # outstanding at the loop's entry, as at every iteration's
    return L


def fixup(blk, drop, site, kh):
    """the guard of `site` tripped for block blk: S (intact, relative to the old reference) -> move the reference by
    max(0, row maximum), rescale l and O, redo the tile's probabilities, return to the site"""
    L = ["o2af_fix%d_%%=:" % site, "s_nop 15", "s_nop 3"]
    L += rowmax(T0, S_[blk])
    L += both_halves(T0, T1, "v_max_f32")
    L.append("v_max_f32 %s, 0, %s" % (V(T0), V(T0)))
    for r in range(16):
        L.append("v_sub_f32 %s, %s, %s" % (V(NR[blk] + r), V(NR[blk] + r), V(T0)))
    for r in range(32):
        L.append("v_sub_f32 %s, %s, %s" % (V(S_[blk] + r), V(S_[blk] + r), V(T0)))
    L.append("v_mul_f32 %s, -1.0, %s" % (V(T1), V(T0)))
    L.append("v_exp_f32 %s, %s" % (V(T1), V(T1)))
    L.append("s_nop 0")
    L.append("v_mul_f32 %s, %s, %s" % (V(L_[blk]), V(L_[blk]), V(T1)))
    for r in range(64):
        L.append("v_accvgpr_read_b32 %s, %s" % (V(T2), A(O_[blk] + r)))
        L.append("v_mul_f32 %s, %s, %s" % (V(T2), V(T2), V(T1)))
        L.append("v_accvgpr_write_b32 %s, %s" % (A(O_[blk] + r), V(T2)))
    L += softmax_stream(blk, drop, kh)
    L.append("s_nop 3")
    L.append("s_branch o2af_ret%d_%%=" % site)
    return L


def epilogue(drop):
    L = []
    e = L.append
    e("s_nop 15")
    e("s_nop 3")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")                                                   # ring free: it becomes the output staging area
    # lane-constant addresses: write [row = lane & 31][chunk (4 db + g4) ^ (row & 15)] + 8 h into the wave's 16 KiB
    e("v_and_b32 %s, 31, %s" % (V(T0), V(VLANE)))
    e("v_lshrrev_b32 %s, 5, %s" % (V(T1), V(VLANE)))
    e("v_lshlrev_b32 %s, 8, %s" % (V(24), V(T0)))
    e("v_lshl_add_u32 %s, %s, 3, %s" % (V(24), V(T1), V(24)))
    e("s_lshl_b32 %s, %%[wave], 14" % S(S_X))
    e("s_add_u32 %s, %s, %s" % (S(S_X), S(S_X), S(S_LDS)))
    e("v_add_u32 %s, %s, %s" % (V(24), S(S_X), V(24)))               # write base
    e("v_and_b32 %s, 15, %s" % (V(T2), V(VLANE)))
    e("v_lshlrev_b32 %s, 4, %s" % (V(25), V(T2)))                    # (row & 15) << 4
    for c in range(16):
        e("v_xor_b32 %s, 0x%x, %s" % (V(32 + c), c << 4, V(25)))
        e("v_add_u32 %s, %s, %s" % (V(32 + c), V(32 + c), V(24)))
    for blk in "XY":
        # l over both halves, 1 / l, lse
        lr = L_[blk]
        L.extend(both_halves(lr, T3, "v_add_f32"))
        e("v_rcp_f32 %s, %s" % (V(T0), V(lr)))
        e("v_log_f32 %s, %s" % (V(T1), V(lr)))
        e("s_nop 0")
        e("v_mul_f32 %s, %%[dscale], %s" % (V(T0), V(T0)))
        e("v_sub_f32 %s, %s, %s" % (V(T1), V(T1), V(NR[blk])))
        e("v_mul_f32 %s, 0x3f317218, %s" % (V(26 if blk == "X" else 27), V(T1)))     # lse = (ref + log2 l) ln 2
        for db in range(4):
            for g4 in range(4):
                base = O_[blk] + 16 * db + 4 * g4
                for k in range(4):
                    e("v_accvgpr_read_b32 %s, %s" % (V(48 + k), A(base + k)))
                for k in range(4):
                    e("v_mul_f32 %s, %s, %s" % (V(48 + k), V(48 + k), V(T0)))
                e("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(52), V(48), V(49)))
                e("v_cvt_pk_bf16_f32 %s, %s, %s" % (V(53), V(50), V(51)))
                e("ds_write_b64 %s, %s offset:%d" % (V(32 + 4 * db + g4), V(52, 2), (8192 if blk == "Y" else 0)))
    # lse: lanes 0..31 store 4 bytes each
    e("v_lshlrev_b32 %s, 2, %s" % (V(T2), V(VLANE)))
    e("s_mov_b64 %s, exec" % S(S_EX, 2))
    e("s_mov_b32 exec_lo, -1")
    e("s_mov_b32 exec_hi, 0")
    e("global_store_dword %s, %s, %%[lptr]" % (V(T2), V(26)))
    e("global_store_dword %s, %s, %%[lptr] offset:128" % (V(T2), V(27)))
    e("s_mov_b64 exec, %s" % S(S_EX, 2))
    # read back whole rows: lane reads row 4 i + (lane >> 4), chunk (lane & 15) ^ (row & 15)
    e("v_lshrrev_b32 %s, 4, %s" % (V(T0), V(VLANE)))                 # lane >> 4
    e("v_and_b32 %s, 15, %s" % (V(T1), V(VLANE)))
    e("v_xor_b32 %s, %s, %s" % (V(T2), V(T1), V(T0)))
    e("v_lshlrev_b32 %s, 4, %s" % (V(T2), V(T2)))
    e("v_lshl_add_u32 %s, %s, 8, %s" % (V(T2), V(T0), V(T2)))
    e("v_add_u32 %s, %s, %s" % (V(T2), S(S_X), V(T2)))
    for k in range(4):
        e("v_xor_b32 %s, 0x%x, %s" % (V(28 + k), (4 * k) << 4, V(T2)))
    e("v_mul_lo_u32 %s, %s, %%[orowb]" % (V(T3), V(T0)))             # (lane >> 4) x row pitch of out
    e("v_lshl_add_u32 %s, %s, 4, %s" % (V(T3), V(T1), V(T3)))
    e("s_waitcnt lgkmcnt(0)")
    e("s_mov_b64 %s, %%[optr]" % S(S_OP, 2))
    e("s_lshl_b32 %s, %%[orowb], 2" % S(S_Y2))                       # 4 rows
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


BASE = dict(klag=1, vlag=1, bar=6, dstride=3, valu_first=0, valu_last=31)


def to1616(lines):
    """TIMING ONLY (cfg abl_1616; results are garbage): every v_mfma_f32_32x32x16_bf16 becomes two v_mfma_f32_16x16x32_bf16 of the same
    FLOPs on the first eight registers of its accumulator block, same operand registers, same place in the stream, and the
    softmax guard never branches -- what this schedule would run at on the other MFMA shape (profiles/r04_attn_1616_probe.txt)"""
    import re
    out = []
    for l in lines:
        m = re.match(r"v_mfma_f32_32x32x16_bf16 ([av])\[(\d+):\d+\], (\S+), (\S+), (?:([av])\[(\d+):\d+\]|0)$", l)
        if m:
            dk, d0, a, b, ck = m.group(1), int(m.group(2)), m.group(3), m.group(4), m.group(5)
            for h in range(2):
                c = "0" if ck is None else "%s[%d:%d]" % (ck, int(m.group(6)) + 4 * h, int(m.group(6)) + 4 * h + 3)
                out.append("v_mfma_f32_16x16x32_bf16 %s[%d:%d], %s, %s, %s" % (dk, d0 + 4 * h, d0 + 4 * h + 3, a, b, c))
        elif l.startswith("v_mfma"):
            raise ValueError("to1616: " + l)
        elif l.startswith("s_cbranch_vccnz o2af_fix") or l.startswith("s_cbranch_vccnz o2dq_fix"):
            out.append("s_nop 0")
        else:
            out.append(l)
    return out


def gen(drop, cfg=None):
    cfg = BASE if cfg is None else cfg
    if cfg.get("abl_1616") and not cfg.get("_in1616"):
        return to1616(gen(drop, dict(cfg, _in1616=1)))
    L = prologue(drop)
    L.append("o2af_loop_%=:")
    body, fix = [], []
    for par in range(2):
        body += phase(1, par, drop, 2 * par, cfg) + phase(2, par, drop, 2 * par + 1, cfg)
        fix += fixup("X", drop, 2 * par, KH[par]) + fixup("Y", drop, 2 * par + 1, KH[par])
    carry = [regs_of(V(KH[0], 4)), regs_of(V(KH[0] + 4, 4))] if drop else []
    body, pend = insert_lgkm_waits(body, carry)
    assert pend == carry, "the LDS reads outstanding at the end of the body differ from those at its entry"
    L += body
    L += ["s_cmp_lt_u32 %s, %%[nt]" % S(S_T), "s_cbranch_scc1 o2af_loop_%="]
    # tail: O_Y += V(nt-1)^T P_Y(nt-1)^T (the fragments are resident)
    L += [x for x, _ in pv_chain("Y")]
    L += ["s_branch o2af_epi_%="]
    L += fix
    L += ["o2af_epi_%=:"]
    L += epilogue(drop)
    return L


def emit(path):
    out = ["// GENERATED by tools/gen_attn_fwd.py -- do not edit; the schedule lives in that script.", "#pragma once",
           "#define O2_AF_KH_OFF %d" % KH_OFF, "#define O2_AF_LDS_BYTES(L) (%d + (L) + 64)" % KH_OFF]

    def macro(name, lines):
        out.append("#define %s \\" % name)
        for k, s in enumerate(lines):
            if s.endswith(":"):
                out.append('  "%s\\n"%s' % (s, " \\" if k + 1 < len(lines) else ""))
            else:
                out.append('  "%s\\n\\t"%s' % (s, " \\" if k + 1 < len(lines) else ""))

    macro("O2_AF_ASM_DROP", gen(True))
    macro("O2_AF_ASM_NODROP", gen(False))
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % r for r in range(256)] + ['"v%d"' % r for r in range(8, 256)] + \
           ['"s%d"' % r for r in range(36, 70)]
    out.append("#define O2_AF_CLOBBERS \\")
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
        out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(root, "orbit-2_amd", "csrc", "attn_fwd_asm.h")
        emit(out)
        print("wrote %s" % out, BASE)
