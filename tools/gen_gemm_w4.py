#!/usr/bin/env python3
"""Writes orbit-2_amd/csrc/gemm_w4_asm.h: the instruction stream of gemm256w_kernel's main loop (csrc/gemm.hip), one
`asm volatile` statement per operand form with every instruction placed by this table.

Shape of the loop (gfx950, one wave per SIMD, 4 waves = one 256 x 256 tile, a wave = 128 x 128 of it):
  * accumulators: a[0:255] = 8 x 8 blocks of v_mfma_f32_16x16x32_bf16, block (i, j) at a[(8 i + j) 4 ...]; the m-side
    fragment is srcB and the n-side fragment srcA, so a lane owns 4 consecutive n (csrc/gemm.hip, epilogue).
  * fragments: two register sets of 8 m-side + 8 n-side fragments (4 registers each): X (k-step 0 of a 64-deep K-tile) =
    v[128:191], Y (k-step 1) = v[192:255].  A K-contiguous operand's fragment is one ds_read_b128 (image [128 rows][64 k],
    16-byte chunk c of row u at c ^ (u & 7): two address registers, one per k-step); a K-strided operand's fragment is two
    ds_read_b64_tr_b16 (image of 16 pieces x 4 k-rows x 256 B with 32 B between pieces, k-row 32 kk + 8 g + 4 s + q in piece
    8 kk + 4 (g & 1) + q at row 2 (g >> 1) + s: every fragment address is the lane's base + an immediate, no column
    swizzle, reads and LDS-DMA writes conflict-free).
  * LDS: 2 stages x 4 units (A rows 0-127, A rows 128-255, B columns 0-127, B columns 128-255) of UNIT bytes.
  * a K-tile = 128 MFMAs ("slots"); the instructions between two MFMAs ("gap g" follows slot g) are placed here:
      gaps 0 ..  : Y reads of K-tile t (stage t & 1), the B operand's first
      bar0       : the B reads have returned in every wave -> the stage's B units are free: B pieces of K-tile t + 2 follow,
                   one every `lstride` gaps (an LDS-DMA piece that misses L2 holds the wave's issue; one wave per SIMD
                   has nobody to cover for it: bursts cost 60-90 cycles per piece, this spacing ~10)
      bar1       : every wave has read all of K-tile t; read bases flip stage; A pieces of K-tile t + 2 follow
      bar2       : vmcnt(pieces of t + 2 issued so far) + s_barrier: K-tile t + 1 has landed for every wave
      then       : X reads of K-tile t + 1; counted lgkmcnt in front of their first uses in the next K-tile
  * K-tiles past the end re-load the last one into a stage nobody reads again: constant vmcnt arithmetic, no branches.
Operands of the statement are named (%[...]); scratch SGPRs s64-s99 and every register above are clobbers."""
import os

UNIT = 16896                       # 16 pieces x (1024 + 32): the K-strided image; the K-contiguous image uses 16384 of it
STAGE = 4 * UNIT
XA = lambda i: 128 + 4 * i
XB = lambda j: 160 + 4 * j
YA = lambda i: 192 + 4 * i
YB = lambda j: 224 + 4 * j
V4 = lambda b: "v[%d:%d]" % (b, b + 3)
V2 = lambda b: "v[%d:%d]" % (b, b + 1)
A4 = lambda b: "a[%d:%d]" % (b, b + 3)
S_PA, S_PB, S_WA, S_T, S_X, S_M0, S_X2, S_WB, S_TA, S_TB, S_DA, S_DB = 64, 72, 80, 81, 82, 83, 84, 85, 86, 87, 88, 92


def frag_reads(op, kc, regs, kk):
    """instructions that fill the 8 fragments `regs` (base register numbers) of operand op ('a' / 'b') for k-step kk"""
    out = []
    for f, r in enumerate(regs):
        if kc:
            out.append(("ds_read_b128 %s, %%[r%s%d] offset:%d" % (V4(r), op, kk, f * 2048), r))
        else:
            off = kk * 8 * 1056 + f * 32
            out.append(("ds_read_b64_tr_b16 %s, %%[r%s0] offset:%d" % (V2(r), op, off), r))
            out.append(("ds_read_b64_tr_b16 %s, %%[r%s0] offset:%d" % (V2(r + 2), op, off + 256), r))
    return out


def tile_loads(a_kc, b_kc):
    """(set-M0, LDS-DMA, is-B) of the 16 pieces a wave stages per K-tile, B units first"""
    out = []
    for which in (2, 3, 0, 1):
        h = which & 1
        for t in range(4):
            if which < 2:
                setm0 = "s_add_u32 m0, s%d, %d" % (S_WA, h * UNIT + t * (1024 if a_kc else 1056))
                ld = "buffer_load_dwordx4 %%[voa], s[%d:%d], s%d offen lds" % (S_DA, S_DA + 3, S_PA + h * 4 + t)
            else:
                setm0 = "s_add_u32 m0, s%d, %d" % (S_WB, h * UNIT + t * (1024 if b_kc else 1056))
                ld = "buffer_load_dwordx4 %%[vob], s[%d:%d], s%d offen lds" % (S_DB, S_DB + 3, S_PB + h * 4 + t)
            out.append((setm0, ld, which >= 2))
    return out


def k_advance(first, kwrap=0):
    """the K advance moves the 48-bit BASE of the two buffer descriptors (SALU): the 32-bit per-lane offset of a K-strided
    operand would pass 2^31 -- the descriptors' range -- at K x row pitch x 2 B > 2 GiB (131072 tokens x 9216 columns).
    kwrap = n (a power of two; TIMING-ONLY ablation, wrong products): the loop walks the first n K-tiles of its panels over and
    over (every n-th advance steps back by n - 1), so the same MFMAs / LDS traffic / LDS-DMA pieces run with the panel reads
    served from L2 -- what the traffic beyond L2 costs (profiles/r06_dw_traffic_clock.txt)."""
    if kwrap:
        n = kwrap
        return first + [
            "s_and_b32 s%d, s%d, %d" % (S_X, S_X, n - 1),
            "s_mul_i32 s96, %%[ka], %d" % -(n - 1),
            "s_mul_i32 s97, %%[kb], %d" % -(n - 1),
            "s_cmp_eq_u32 s%d, 0" % S_X,
            "s_cselect_b32 s%d, s96, %%[ka]" % S_X,
            "s_cselect_b32 s%d, s97, %%[kb]" % S_X2,
            "s_ashr_i32 s96, s%d, 31" % S_X,
            "s_ashr_i32 s97, s%d, 31" % S_X2,
            "s_add_u32 s%d, s%d, s%d" % (S_DA, S_DA, S_X),
            "s_addc_u32 s%d, s%d, s96" % (S_DA + 1, S_DA + 1),
            "s_add_u32 s%d, s%d, s%d" % (S_DB, S_DB, S_X2),
            "s_addc_u32 s%d, s%d, s97" % (S_DB + 1, S_DB + 1),
        ]
    return first + [
        "s_cmp_lt_u32 s%d, %%[nk]" % S_X,
        "s_cselect_b32 s%d, %%[kb], 0" % S_X2,
        "s_cselect_b32 s%d, %%[ka], 0" % S_X,
        "s_add_u32 s%d, s%d, s%d" % (S_DA, S_DA, S_X),
        "s_addc_u32 s%d, s%d, 0" % (S_DA + 1, S_DA + 1),
        "s_add_u32 s%d, s%d, s%d" % (S_DB, S_DB, S_X2),
        "s_addc_u32 s%d, s%d, 0" % (S_DB + 1, S_DB + 1),
    ]


def flip_write_bases():
    return ["s_sub_u32 s%d, s%d, s%d" % (S_WA, S_TA, S_WA), "s_sub_u32 s%d, s%d, s%d" % (S_WB, S_TB, S_WB)]


def gen(a_kc, b_kc, cfg):
    L = []
    e = L.append
    e("s_nop 4")
    e("s_mov_b32 s%d, m0" % S_M0)
    for (S, p0, ps, ph) in ((S_PA, "%[pa0]", "%[psa]", "%[pha]"), (S_PB, "%[pb0]", "%[psb]", "%[phb]")):
        e("s_mov_b32 s%d, %s" % (S, p0))                      # piece offsets of this wave: unit half h, piece t -> s[S + 4 h + t]
        for t in range(1, 4):
            e("s_add_u32 s%d, s%d, %s" % (S + t, S + t - 1, ps))
        for t in range(4):
            e("s_add_u32 s%d, s%d, %s" % (S + 4 + t, S + t, ph))
    for S, base in ((S_DA, "%[abase]"), (S_DB, "%[bbase]")):          # raw buffer descriptors: base, stride 0, 2^31 - 1 bytes
        e("s_mov_b64 s[%d:%d], %s" % (S, S + 1, base))
        e("s_mov_b32 s%d, 0x7fffffff" % (S + 2))
        e("s_mov_b32 s%d, 0x00020000" % (S + 3))
    if cfg.get("pace"):
        e("s_mov_b32 s97, 0")
    e("s_mov_b32 s%d, %%[ldswa]" % S_WA)
    e("s_mov_b32 s%d, %%[ldswb]" % S_WB)
    e("s_lshl_b32 s%d, s%d, 1" % (S_TA, S_WA))               # stage flip of a write base: base <- (2 base0 + STAGE) - base
    e("s_add_u32 s%d, s%d, %d" % (S_TA, S_TA, STAGE))
    e("s_lshl_b32 s%d, s%d, 1" % (S_TB, S_WB))
    e("s_add_u32 s%d, s%d, %d" % (S_TB, S_TB, STAGE))
    loads = tile_loads(a_kc, b_kc)
    for tile in range(2):                                     # K-tiles 0 and 1 -> stages 0 and 1
        for setm0, ld, _ in loads:
            e(setm0)
            e("s_nop 0")
            e(ld)
        for x in k_advance(["s_mov_b32 s%d, %d" % (S_X, tile + 1)], cfg.get("kwrap", 0)):
            e(x)
        for x in flip_write_bases():
            e(x)
    for r in range(256):
        e("v_accvgpr_write_b32 a%d, 0" % r)
    e("s_waitcnt vmcnt(16)")
    e("s_barrier")
    xreads = frag_reads("b", b_kc, [XB(j) for j in range(8)], 0) + frag_reads("a", a_kc, [XA(i) for i in range(8)], 0)
    yb = frag_reads("b", b_kc, [YB(j) for j in range(8)], 1)
    ya = frag_reads("a", a_kc, [YA(i) for i in range(8)], 1)
    for ins, _ in xreads:
        e(ins)
    e("s_mov_b32 s%d, 0" % S_T)
    if cfg.get("stamp"):
        e("s_waitcnt lgkmcnt(0)")
    e("o2w4_loop_%=:")

    gaps = [[] for _ in range(128)]
    lds_issue = []                               # (gap, destination fragment) of every Y read of the K-tile, in issue order
    g = 0
    for ins, r in yb:
        gaps[g].append(ins); lds_issue.append((g, r)); g += (2 if b_kc else 1)
    last_b = g - (2 if b_kc else 1)
    for ins, r in ya:
        gaps[g].append(ins); lds_issue.append((g, r)); g += (2 if a_kc else 1)
    last_y = max(x for x, _ in lds_issue)
    b0 = last_b + 3
    n_after = sum(1 for x, _ in lds_issue[len(yb):] if x <= b0)
    assert n_after <= 15
    gaps[b0] += ["s_waitcnt lgkmcnt(%d)" % n_after, "s_barrier"]
    b1 = max(last_y + cfg["bar1_lag"], b0 + 2)
    assert b1 < 62
    gaps[b1] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    flips = ["v_sub_u32 %%[r%s], %%[t%s], %%[r%s]" % (r, r, r) for r in
             (["a0", "a1"] if a_kc else ["a0"]) + (["b0", "b1"] if b_kc else ["b0"])]
    for k, ins in enumerate(flips):
        gaps[b1 + 1 + k // 2].append(ins)
    # LDS-DMA pieces of K-tile t + 2: B pieces from bar0, A pieces from bar1
    lg, nxt = [], b0 + 2
    for k, (setm0, ld, is_b) in enumerate(loads):
        if not is_b:
            nxt = max(nxt, b1 + 4)
        lg.append(nxt)
        nxt += cfg["lstride"]
    for (setm0, ld, _), g_ in zip(loads, lg):
        gaps[g_ - 1].append(setm0)
        gaps[g_].append(ld)
    b2 = cfg["bar2"]
    if cfg.get("pace"):
        # cohort pacing (the weight-gradient form; csrc/gemm.hip w4_cohort_start): every P-th K-tile, in front of bar2, the wave that
        # holds the cohort's counter (%[pcp] != 0: wave 0 of a paced launch) adds its arrival and polls -- bounded -- until the
        # whole cohort (the XCD's workgroups of this round, %[pcn] of them) has arrived; the other waves wait at bar2's s_barrier.
        # A cohort's column groups sweep at rates 3-5 % apart (memory-side); L2 holds ~10 K-tiles of the cohort's strips: a
        # check every P = 64 K-tiles keeps them within 2-3.  v128 / v129 are dead here (the X fragments of K-tile t were used
        # up in slots 0-63 and those of t + 1 are read after bar2); s96-s99 are scratch (s97 = the running target, s[98:99] = exec).
        P = cfg["pace"]
        gaps[b2] += [
            "s_and_b32 s96, s%d, %d" % (S_T, P - 1),
            "s_cmp_eq_u32 s96, %d" % (P - 1),
            "s_cbranch_scc0 o2w4_np_%=",
            "s_cmp_eq_u64 %[pcp], 0",
            "s_cbranch_scc1 o2w4_np_%=",
            "s_mov_b64 s[98:99], exec",
            "s_mov_b64 exec, 1",                               # one lane: one arrival, one polled dword
            "v_mov_b32 v129, 0",
            "v_mov_b32 v128, 1",
            "global_atomic_add v129, v128, %[pcp] sc1",
            "s_add_u32 s97, s97, %[pcn]",
            "s_mov_b32 s%d, %d" % (S_X, cfg.get("pace_polls", 24)),     # (S_X / S_X2 are free outside the K-advance gaps)
            "o2w4_pl_%=:",
            "global_load_dword v128, v129, %[pcp] sc1",
            "s_waitcnt vmcnt(0)",
            "v_readfirstlane_b32 s%d, v128" % S_X2,
            "s_nop 0",
            "s_cmp_ge_u32 s%d, s97" % S_X2,
            "s_cbranch_scc1 o2w4_pd_%=",
            "s_sleep 4",
            "s_sub_u32 s%d, s%d, 1" % (S_X, S_X),
            "s_cmp_lg_u32 s%d, 0" % S_X,
            "s_cbranch_scc1 o2w4_pl_%=",
            "o2w4_pd_%=:",
            "s_mov_b64 exec, s[98:99]",
            "o2w4_np_%=:",
        ]
    gaps[b2] += ["s_waitcnt vmcnt(%d)" % sum(1 for x in lg if x < b2), "s_barrier"]
    ka_ = k_advance(["s_add_u32 s%d, s%d, 3" % (S_X, S_T)], cfg.get("kwrap", 0))
    ga = max(lg) + 1
    assert ga + 3 <= 125, "pieces run too late"
    assert not cfg.get("pace") or not (ga <= b2 <= ga + 2), "the pacing code borrows S_X / S_X2: keep it out of the K-advance gaps"
    if cfg.get("kwrap"):
        gaps[ga] += ka_[:7]                      # (ablation) s_cmp + s_cselects stay together (SCC)
        gaps[ga + 1] += ka_[7:11]                # s_add / s_addc pairs stay together (carry)
        gaps[ga + 2] += ka_[11:13]
    else:
        gaps[ga] += ka_[:4]                      # s_cmp + s_cselects stay together (SCC)
        gaps[ga + 1] += ka_[4:6]                 # s_add / s_addc pairs stay together (carry)
        gaps[ga + 2] += ka_[6:8]
    gaps[ga + 3] += flip_write_bases()
    # X reads of K-tile t + 1
    g = b2 + 1
    xissue = []
    for ins, r in xreads:
        assert g <= 125, "X reads do not fit"
        gaps[g].append(ins); xissue.append((g, r))
        g += cfg["xstride_kc"] if ins.startswith("ds_read_b128") else 1
    gaps[126] += ["s_add_u32 s%d, s%d, 1" % (S_T, S_T)]
    # counted waits in front of the first use of each X fragment (LDS reads return in order; the Y reads of this K-tile that
    # were issued before the slot are younger than every X read)
    head = {}
    if not cfg.get("stamp"):
        first_use = {}
        for m in range(64):
            i, j = m // 8, m % 8
            for r in (XA(i), XB(j)):
                first_use.setdefault(r, m)
        by_slot = {}
        for r, m in first_use.items():
            by_slot.setdefault(m, []).append(r)
        done = -1                                # X reads up to this issue index are known complete
        for m in sorted(by_slot):
            last_needed = max(k for k, (_, r) in enumerate(xissue) if r in by_slot[m])
            if last_needed <= done or m > b1:    # (bar1's lgkmcnt(0) covers every X read)
                continue
            outstanding = (len(xissue) - 1 - last_needed) + sum(1 for x, _ in lds_issue if x < m)
            n = min(15, outstanding)
            head[m] = "s_waitcnt lgkmcnt(%d)" % n
            done = last_needed + (outstanding - n)   # a clamped count retires younger reads too
    if cfg.get("stamp"):
        # s[96:97] = the previous stamp, s[98:99] = this one; each point adds its distance to the previous one to %[tK]
        def point(name):
            return ["s_memtime s[98:99]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s98, s96" % S_X,
                    "s_add_u32 %%[%s], %%[%s], s%d" % (name, name, S_X), "s_mov_b32 s96, s98"]
        k1 = next(k for k, x in enumerate(gaps[b1]) if x.startswith("s_waitcnt lgkmcnt"))
        gaps[b1][k1:k1] = point("t0")
        k1 = next(k for k, x in enumerate(gaps[b1]) if x.startswith("s_barrier"))
        gaps[b1][k1 + 1:k1 + 1] = point("t1")
        k2 = next(k for k, x in enumerate(gaps[b2]) if x.startswith("s_waitcnt vmcnt"))
        gaps[b2][k2:k2] = point("t2")
        k2 = next(k for k, x in enumerate(gaps[b2]) if x.startswith("s_barrier"))
        gaps[b2][k2 + 1:k2 + 1] = point("t3")
        L += ["s_memtime s[96:97]", "s_waitcnt lgkmcnt(0)"]
    for m in range(128):
        kk, i, j = m // 64, (m % 64) // 8, m % 8
        fa = (XA if kk == 0 else YA)(i)
        fb = (XB if kk == 0 else YB)(j)
        acc = (i * 8 + j) * 4
        if m in head:
            L.append(head[m])
        L.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (A4(acc), V4(fb), V4(fa), A4(acc)))
        L += gaps[m]
    if cfg.get("stamp"):
        for x in ["s_memtime s[98:99]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s98, s96" % S_X,
                  "s_add_u32 %%[t4], %%[t4], s%d" % S_X]:
            e(x)
    e("s_cmp_lt_u32 s%d, %%[nk]" % S_T)
    e("s_cbranch_scc1 o2w4_loop_%=")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if cfg.get("pace"):
        # (after the drain: the last X reads land in v128.. too)  a workgroup that leaves its sweep credits the cohort's counter beyond any target (targets stay below 32 x 2048 / P): in a
        # cohort whose members sweep contractions of different lengths the ones still sweeping then pass every later check at once
        # instead of waiting out the poll budget for a member that is gone
        L += ["s_cmp_eq_u64 %[pcp], 0", "s_cbranch_scc1 o2w4_nc_%=",
              "s_mov_b64 s[98:99], exec", "s_mov_b64 exec, 1",
              "v_mov_b32 v129, 0", "v_mov_b32 v128, 0x100000",
              "global_atomic_add v129, v128, %[pcp] sc1",
              "s_mov_b64 exec, s[98:99]", "o2w4_nc_%=:"]
    e("s_nop 7")
    e("s_nop 7")
    e("s_mov_b32 m0, s%d" % S_M0)
    return L


def gen_cstage(hh):
    """accumulator blocks (i = 4 hh + ii, j) -> the fp32 staging image of the epilogue (csrc/gemm.hip, w4_epilogue_rows):
    ds_write_b128 straight from the accumulator registers; %[be] / %[bo] = the lane's address for even / odd j"""
    L = []
    for ii in range(4):
        for j in range(8):
            acc = ((hh * 4 + ii) * 8 + j) * 4
            L.append("ds_write_b128 %%[%s], %s offset:%d" % ("bo" if j & 1 else "be", A4(acc), ii * 16384 + (j >> 1) * 128))
    L.append("s_waitcnt lgkmcnt(0)")
    return L


BASE = dict(bar1_lag=6, lstride=6, bar2=90, xstride_kc=2, pace=64)
FORMS = {"NT": (True, True), "NN": (True, False), "TN": (False, False), "TT": (False, True)}


def emit(path):
    out = ["// GENERATED by tools/gen_gemm_w4.py -- do not edit; the schedule lives in that script.", "#pragma once",
           "#define O2_W4_UNIT %d" % UNIT]

    def macro(name, lines):
        out.append("#define %s \\" % name)
        for k, s in enumerate(lines):
            out.append('  "%s\\n\\t"%s' % (s, " \\" if k + 1 < len(lines) else ""))

    for hh in range(2):
        macro("O2_W4_CSTAGE%d" % hh, gen_cstage(hh))
    for name, (a_kc, b_kc) in FORMS.items():
        cfg = dict(BASE)
        if name != "TN" or cfg.get("kwrap") or cfg.get("kwrap_tn"):
            cfg.pop("pace", None)                # (pacing lives in the weight-gradient form only; the wrap ablation borrows its scratch
                                                 #  registers s96 / s97, so the two never share a loop)
        if cfg.get("kwrap_tn"):                  # (ablation restricted to the weight-gradient form: a step's activations stay right)
            cfg["kwrap"] = cfg["kwrap_tn"] if name == "TN" else 0
        macro("O2_W4_ASM_%s" % name, gen(a_kc, b_kc, cfg))
        macro("O2_W4_ASM_%s_STAMP" % name, gen(a_kc, b_kc, dict(BASE, stamp=True, pace=0)))
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"a%d"' % r for r in range(256)] + ['"v%d"' % r for r in range(128, 256)] + \
           ['"s%d"' % r for r in range(64, 100)]
    out.append("#define O2_W4_CLOBBERS \\")
    for k in range(0, len(clob), 16):
        chunk = ", ".join(clob[k:k + 16])
        out.append("  %s%s" % (chunk, ", \\" if k + 16 < len(clob) else ""))
    open(path, "w").write("\n".join(out) + "\n")


def show(name):
    a_kc, b_kc = FORMS[name]
    lines = gen(a_kc, b_kc, BASE)
    slot = -1
    for l in lines[lines.index("o2w4_loop_%=:") + 1:]:
        if l.startswith("v_mfma"):
            slot += 1
            continue
        print(slot, l)


if __name__ == "__main__":
    import sys
    # tuning: --cfg "lstride=5,bar2=88" overrides BASE, --out PATH writes the header elsewhere (tools/mkvar_w4.sh)
    if "--cfg" in sys.argv:
        for kv in sys.argv[sys.argv.index("--cfg") + 1].split(","):
            k, v = kv.split("=")
            BASE[k] = int(v)
    if len(sys.argv) > 2 and sys.argv[1] == "show":
        show(sys.argv[2])
    else:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(root, "orbit-2_amd", "csrc", "gemm_w4_asm.h")
        emit(out)
        print("wrote %s:" % out, ", ".join(FORMS), BASE)
