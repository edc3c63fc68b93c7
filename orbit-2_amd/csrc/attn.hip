// Flash-style multi-head self-attention for gfx950: forward, and a two-kernel backward
// (dQ: query on the MFMA lane, loop over keys;  dK/dV: key on the lane, loop over queries --
// no cross-workgroup reduction, no atomics, bitwise reproducible).
//
// Layout: qkv is the qkv-Linear output as stored, [B, L, 3, H, D] bf16 (reference: attention.py:50
// reshapes+permutes it; here the kernels read it strided, so no permute copies exist); out is
// [B, L, H, D] = the [B*L, C] operand of the proj GEMM.
//
// All products use v_mfma_f32_32x32x16_bf16.  A 32x32 f32 accumulator X (column on the lane, rows in the
// 16 registers) is fed to the next MFMA as an operand without touching LDS: registers 8s..8s+7 -> bf16 is
// the fragment of k-step s, whose k order is row(j,h) = 16s + 8(j>>2) + 4h + (j&3); the other operand is
// fetched in that same order with ds_read_b64_tr_b16 (two 4-row transposed blocks).
//
// LDS images are [row][D] with one XOR swizzle that is conflict-free for both ds_read_b128 row reads and
// the transposed reads (16-byte chunk c of row r stored at chunk c ^ f(r)); tiles arrive by LDS-DMA with
// the swizzle applied on the per-lane SOURCE address (destination is lane-linear).
#include "tiles32.h"
#ifndef O2_DKV256_LA
#define O2_DKV256_LA 2      /* k-steps of LDS operands in flight in the d = 256 fused dK+dV kernel */
#endif
#include "../../include/orbit2_hip.h"
#include "attn_fwd_asm.h"
#include "attn_dq_asm.h"
#include "attn_dkv_asm.h"

namespace {

#ifdef O2_STAMP
// Diagnostic build only (tools/stamp_build.sh): wave 0 of the first 64 workgroups of the forward / dK kernels sums the shader
// cycles of each segment of its tile loop; nothing in the kernels reads these words.
__device__ unsigned int o2_dbg_attn[64 * 8];
#define O2_T() ((unsigned)__builtin_amdgcn_s_memtime())
#define O2_SEG(acc) { t1_ = O2_T(); acc += t1_ - t0_; t0_ = t1_; }
#else
#define O2_SEG(acc)
#endif

__device__ __forceinline__ bf16x8 pack_frag(const f32x16& x, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(x[8 * s + j]);
  return r;
}

// two floats -> packed bf16 pair as ONE v_cvt_pk_bf16_f32 (the shift-and-or form of pack_bf2 costs three more instructions)
__device__ __forceinline__ unsigned cvt_pk_bf2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// bf16 fragment times a scalar, rounded to bf16 once (the softmax scale folded into an MFMA operand)
__device__ __forceinline__ bf16x8 scale_frag(const bf16x8& x, float c) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(bf2f((bf16_t)x[j]) * c);
  return r;
}

template <int J>
__device__ __forceinline__ uint32_t quad_bcast_c(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J * 0x55, 0xf, 0xf, true);
}
// value of lane (4*(lane/4) + j); j is a compile-time constant after unrolling
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v, int j) {
  switch (j) {
    case 0: return quad_bcast_c<0>(v);
    case 1: return quad_bcast_c<1>(v);
    case 2: return quad_bcast_c<2>(v);
    default: return quad_bcast_c<3>(v);
  }
}


// dropout on a 32x32 tile whose lane-local axis (registers) runs along KEYS: registers 4t..4t+3 are keys
// kbase + 8t + 4h + {0..3} of query row qrow -> one hash per 4 registers.
// kh4[t] = key-group hash of keys kbase + 8t + 4h + {0..3} (from the tile's LDS table, see stage_keyhash)
// PIN (forward, where the masked values are converted to bf16 next): the select stays in fp32 behind an opaque asm --
// hipcc otherwise converts each value alone, selects, and merges pairs by v_perm (+3.6 % on the d = 128 forward); in the
// dQ kernel the masked values feed fp32 arithmetic and the pin costs 2 %.
template <bool PIN>
__device__ __forceinline__ void drop_keys_in_regs(f32x16& p, uint32_t rowhash, const u32x4& kh4, unsigned thr,
                                                  float dropped = 0.f) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t hh = o2_attn_mix(rowhash, kh4[t]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = (((hh >> (8 * e)) & 0xffu) >= thr) ? p[4 * t + e] : dropped;
      if (PIN) asm("" : "+v"(v));
      p[4 * t + e] = v;
    }
  }
}

// Key-group hashes of one 64-key tile (16 groups), written by 16 lanes of wave 0 one tile ahead.  Entry order:
// group i = 2j + h (j = kb*4 + t) is stored at h*8 + j, so half-wave h reads its 8 values as two 16-byte pieces.
__device__ __forceinline__ void stage_keyhash(uint32_t* skh, uint64_t seed, int tile, int tid) {
  if (tid < 16) skh[(tid & 1) * 8 + (tid >> 1)] = o2_attn_keyhash(seed, (uint32_t)(tile * 16 + tid));
}

// Workgroup -> (row tile of NW*32 rows, head, batch).  The grid is launched one-dimensional; hardware deals consecutive
// workgroups round-robin to the 8 XCDs, so the id is first remapped (bijectively, any grid size) to give every XCD a
// CONTIGUOUS range of tile ids: the ~64 workgroups an XCD runs at a time are then consecutive tiles of one or two
// (batch, head) pairs and share that pair's K/V (or Q/dO) through the XCD's L2, instead of every XCD streaming the
// tiles of eight different heads.
__device__ __forceinline__ void attn_tile_coords(int nq, int H, int& qt, int& head, int& b) {
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  qt = id % nq;
  const int grp = id / nq;
  head = grp % H;
  b = grp / H;
}

// =============================================================================================
// forward
// =============================================================================================
// Measured and not kept (round 2, profiles/r02_attn_fwd_pingpong_*.txt): a software-pipelined ping-pong form of this loop
// (fragments preloaded into registers, pure 16-MFMA segments, waves 4-7 one segment behind waves 0-3, inline-asm LDS-DMA
// rings) is bit-identical and its MFMA segments run at 500-550 cycles per 16 MFMAs, but a wave's vector-ALU + LDS-issue
// + staging work per 64-key tile (~1650 cycles, ~2300 with dropout) is 1.6-2.2x its 1024 MFMA cycles: strict alternation
// costs 2 x that, the free-running loop below already sits near (vector + MFMA) per wave with the SIMD's two waves
// overlapping each other: -4 ... -7 % at d = 128.  The lever left is the vector instruction count per score element.
//
// NW = waves per workgroup (32 query rows each).  8 waves (256-row tiles, one workgroup per CU) share every K/V tile:
// half the LDS-DMA pieces per wave and per MFMA of the 4-wave / two-workgroups-per-CU form (the stamps of that form,
// profiles/r02_attn_fwd_stamps_4wave.txt, show 8 pieces per wave and tile costing 700-970 of ~3200-5100 cycles: the CU's
// LDS write path is busy ~11 cycles per 1-KiB piece and all 8 resident waves queue on it).
#ifndef O2_FWD256_LA
#define O2_FWD256_LA 3
#endif
template <int D, bool DROP, bool RAGGED, int NW>
__global__ __launch_bounds__(NW * 64, (D == 256 ? 1 : 2)) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                          float* __restrict__ lse, int L, int H, float sc_log2,
                                                          unsigned thr, float dscale, uint64_t seed_arg, int ldo, int ldq) {
  const uint64_t seed = seed_arg ^ o2_seed_salt;   // see common.h: fresh masks for every replay of a captured step
  using C = Cfg<D>;
  __shared__ __attribute__((aligned(16))) char smem[4 * C::TILE];  // [2 stages][K | V]
  __shared__ __attribute__((aligned(16))) uint32_t skh[2][16];      // [stage] key-group hashes of the tile (dropout)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hq = lane >> 5;  // MFMA half
  int tile_i, head, b;
  attn_tile_coords((L + NW * 32 - 1) / (NW * 32), H, tile_i, head, b);
  const int q0 = tile_i * (NW * 32) + wave * 32;
  const size_t tstride = (size_t)ldq;      // token-row pitch of qkv (and dqkv), >= 3 * H * D  // token stride in qkv
  const bf16_t* qbase = qkv + (size_t)b * L * tstride + (size_t)head * D;
  const bf16_t* kbase = qbase + (size_t)H * D;
  const bf16_t* vbase = qbase + (size_t)2 * H * D;
  const int qrow_raw = q0 + (lane & 31);
  const bool q_ok = !RAGGED || qrow_raw < L;
  const int qrow = q_ok ? qrow_raw : L - 1;   // ragged tail: compute on a valid row, never store it

  bf16x8 qf[C::NDS];
#pragma unroll
  for (int ds = 0; ds < C::NDS; ++ds)
    qf[ds] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrow * tstride + ds * 16 + 8 * hq);

  f32x16 o[C::NDB];
#pragma unroll
  for (int i = 0; i < C::NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const uint32_t rowhash = DROP ? o2_attn_rowhash(seed, (uint64_t)(b * H + head) * L + (uint64_t)qrow) : 0u;

  const int nt = (L + 63) / 64;
  stage64<D, RAGGED, NW>(kbase, tstride, smem, wave, lane, L);
  stage64<D, RAGGED, NW>(vbase, tstride, smem + C::TILE, wave, lane, L);
  if (DROP) stage_keyhash(skh[0], seed, 0, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
#ifdef O2_STAMP
  unsigned tS = 0, tQK = 0, tSM = 0, tPV = 0, tW = 0, tB = 0, t0_ = O2_T(), t1_;
  const unsigned tstart_ = t0_;
#endif
  for (int t = 0; t < nt; ++t) {
    const char* sk = smem + cur * 2 * C::TILE;
    const char* sv = sk + C::TILE;
    if (t + 1 < nt) {
      char* nk = smem + (cur ^ 1) * 2 * C::TILE;
      stage64<D, RAGGED, NW>(kbase + (size_t)(t + 1) * 64 * tstride, tstride, nk, wave, lane, L - (t + 1) * 64);
      stage64<D, RAGGED, NW>(vbase + (size_t)(t + 1) * 64 * tstride, tstride, nk + C::TILE, wave, lane, L - (t + 1) * 64);
      if (DROP) stage_keyhash(skh[cur ^ 1], seed, t + 1, tid);
    }
    O2_SEG(tS)
    // S^T[kb] = K_kb . Q^T   (rows = keys in registers, column = query on the lane)
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    // the two key blocks' chains are interleaved: a K fragment is consumed two MFMAs after the previous one of its
    // chain, so its LDS read has twice the time to land
    if constexpr (D == 256) {
      // one wave per SIMD (512 registers): nobody else covers an LDS read's latency, and left alone hipcc puts `s_waitcnt
      // lgkmcnt(0)` between every fragment read and its MFMA (64 exposed LDS latencies per tile: the kernel ran at 0.24 of the
      // MFMA peak).  O2_FWD256_LA fragments in flight ahead of the MFMA that consumes them, the order pinned (the discipline of
      // attn_bwd_dkv256_kernel); same products in the same order: the same bits.
      constexpr int LA = O2_FWD256_LA;      // k-steps ahead: 2 LA fragments (both key blocks of a k-step) in flight
      bf16x8 k0[LA + 1], k1[LA + 1];
#pragma unroll
      for (int i = 0; i < LA; ++i) {
        k0[i] = row_frag<D>(sk, (lane & 31), i, hq);
        k1[i] = row_frag<D>(sk, 32 + (lane & 31), i, hq);
      }
#pragma unroll
      for (int ds = 0; ds < C::NDS; ++ds) {
        if (ds + LA < C::NDS) {
          k0[(ds + LA) % (LA + 1)] = row_frag<D>(sk, (lane & 31), ds + LA, hq);
          k1[(ds + LA) % (LA + 1)] = row_frag<D>(sk, 32 + (lane & 31), ds + LA, hq);
        }
        s[0] = MFMA32(k0[ds % (LA + 1)], qf[ds], s[0]);
        s[1] = MFMA32(k1[ds % (LA + 1)], qf[ds], s[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int ds = 0; ds < C::NDS; ++ds)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          s[kb] = MFMA32(row_frag<D>(sk, kb * 32 + (lane & 31), ds, hq), qf[ds], s[kb]);
    }
    if (RAGGED && t == nt - 1 && (L & 63)) {   // keys past the end of a ragged sequence get no weight
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hq >= L) s[kb][r] = -1e30f;
    }
    O2_SEG(tQK)
    // online softmax over this lane's query row (its 32 keys + the partner half's 32)
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // Deferred rescale: the reference max m_run only moves when some row of this wave outgrows it by more
    // than 2^RESCALE_THR (wave-uniform decision), so the O-wide multiply is skipped on almost every tile.
    // P is then bounded by 2^RESCALE_THR instead of 1; O and l use the same reference, the result is exact.
    constexpr float RESCALE_THR = 5.0f;
    const float mt = mx * sc_log2;
    float alpha = 1.0f;
    if (__any(mt > m_run + RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mt);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < C::NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[kb][r] * sc_log2 - m_run);
        s[kb][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    if (DROP) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        drop_keys_in_regs<true>(s[kb], rowhash, *reinterpret_cast<const u32x4*>(&skh[cur][hq * 8 + kb * 4]), thr);
    }
    O2_SEG(tSM)
    // O^T[db] += V^T . P^T
    if constexpr (D == 256) {
      bf16x8 pf[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) pf[g] = pack_frag(s[g >> 1], g & 1);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int NV = 4 * C::NDB, LA = O2_FWD256_LA;      // fragment i: key group (kb, ss) = i / NDB, d block i % NDB
      bf16x8 vr[LA + 1];
#pragma unroll
      for (int i = 0; i < LA; ++i) vr[i] = tr_frag<D>(sv, (i / C::NDB) * 16, i % C::NDB, lane);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (i + LA < NV) vr[(i + LA) % (LA + 1)] = tr_frag<D>(sv, ((i + LA) / C::NDB) * 16, (i + LA) % C::NDB, lane);
        o[i % C::NDB] = MFMA32(vr[i % (LA + 1)], pf[i / C::NDB], o[i % C::NDB]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 pf = pack_frag(s[kb], ss);
#pragma unroll
          for (int db = 0; db < C::NDB; ++db)
            o[db] = MFMA32(tr_frag<D>(sv, kb * 32 + ss * 16, db, lane), pf, o[db]);
        }
    }
    O2_SEG(tPV)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    O2_SEG(tW)
    __syncthreads();
    O2_SEG(tB)
    cur ^= 1;
  }
#ifdef O2_STAMP
  if (blockIdx.x < 64 && wave == 0 && lane == 0) {
    unsigned* dd_ = o2_dbg_attn + blockIdx.x * 8;
    dd_[0] = tS; dd_[1] = tQK; dd_[2] = tSM; dd_[3] = tPV; dd_[4] = tW; dd_[5] = tB; dd_[6] = O2_T() - tstart_; dd_[7] = (unsigned)nt;
  }
#endif
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = (DROP ? dscale : 1.0f) / l_tot;   // dropout scale folded out of the inner loop
  if (!q_ok) return;
  if (hq == 0) lse[((size_t)(b * H + head)) * L + qrow] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
  bf16_t* orow = out + ((size_t)b * L + qrow) * (size_t)ldo + (size_t)head * D;   // ldo: token-row pitch of out (>= H * D)
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dd = db * 32 + 8 * g4 + 4 * hq;
      u32x2 w;
      w[0] = pack_bf2(o[db][4 * g4] * inv, o[db][4 * g4 + 1] * inv);
      w[1] = pack_bf2(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv);
      *reinterpret_cast<u32x2*>(orow + dd) = w;
    }
}

// =============================================================================================
// forward WITHOUT dropout (eval / inference, attn_drop = 0): lazy reference
// =============================================================================================
// Q enters the S MFMAs pre-multiplied by log2(e)/sqrt(d) and the accumulators START at -(reference of the row): the scores
// leave the matrix pipe as exp2 arguments.  The reference is the row's maximum over the first tile and stays FIXED while no
// probability relative to it leaves [0, 2^40]: P, O and l share it and fp32 / bf16 keep their relative precision at any scale,
// so the result is exact, and a score element costs one exp2, one add and its share of the bf16 conversion (124 vector
// instructions per 64-key tile against 200 in attn_fwd_kernel<.., DROP = false, ..>: +6 % at the interm_1b shape,
// profiles/r03_attn_fwd_variants.txt).  The row sums are the guard: a wave in which one leaves the range continues in an
// online-softmax loop.  With dropout the same formulation is 6-16 % SLOWER than attn_fwd_kernel in five variants (guard
// placement, priority, pinned / free select, subtract instead of accumulator start): hipcc's schedule of that loop, not its
// instruction count (243 against 306), decides -- the dropout forward stays as it was.
template <int D, bool RAGGED, int NW>
__global__ __launch_bounds__(NW * 64, (D == 256 ? 1 : 2)) void attn_fwd_lazy_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                          float* __restrict__ lse, int L, int H, float sc_log2,
                                                          unsigned thr, float dscale, uint64_t seed_arg, int ldo, int ldq) {
  constexpr bool DROP = false;                      // (the dropout forward is attn_fwd_kernel above)
  const uint64_t seed = seed_arg ^ o2_seed_salt;
  using C = Cfg<D>;
  constexpr int KT = 1;      // 64-key tiles per stage and per workgroup barrier (2 was measured: no gain)
  constexpr int STAGE = KT * 2 * C::TILE;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];        // [2 stages][KT][K | V]
  __shared__ __attribute__((aligned(16))) uint32_t skh[2][KT * 16];    // [stage][tile] key-group hashes of the tile (dropout)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hq = lane >> 5;  // MFMA half
  int tile_i, head, b;
  attn_tile_coords((L + NW * 32 - 1) / (NW * 32), H, tile_i, head, b);
  const int q0 = tile_i * (NW * 32) + wave * 32;
  const size_t tstride = (size_t)ldq;      // token-row pitch of qkv (and dqkv), >= 3 * H * D  // token stride in qkv
  const bf16_t* qbase = qkv + (size_t)b * L * tstride + (size_t)head * D;
  const bf16_t* kbase = qbase + (size_t)H * D;
  const bf16_t* vbase = qbase + (size_t)2 * H * D;
  const int qrow_raw = q0 + (lane & 31);
  const bool q_ok = !RAGGED || qrow_raw < L;
  const int qrow = q_ok ? qrow_raw : L - 1;   // ragged tail: compute on a valid row, never store it

  // Q enters the S MFMAs pre-multiplied by log2(e)/sqrt(d) (rounded to bf16 once, here): the scores leave the matrix
  // pipe in the exp2 domain, and with the accumulators STARTED at -m_run they are the exp2 arguments -- no multiply, no
  // subtraction and (below) no row maximum per score element.
  bf16x8 qf[C::NDS];
#pragma unroll
  for (int ds = 0; ds < C::NDS; ++ds)
    qf[ds] = scale_frag(*reinterpret_cast<const bf16x8*>(qbase + (size_t)qrow * tstride + ds * 16 + 8 * hq), sc_log2);

  f32x16 o[C::NDB];
#pragma unroll
  for (int i = 0; i < C::NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = 0.f, l_run = 0.f;
  const uint32_t rowhash = DROP ? o2_attn_rowhash(seed, (uint64_t)(b * H + head) * L + (uint64_t)qrow) : 0u;

  const int nt = (L + 63) / 64;
  int cur = 0;
  // stage the KT tiles of super-tile T into buffer `buf` (tiles past the end of the sequence are not staged, nor read)
  auto stage_super = [&](int T, int buf) {
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const int tt = T * KT + j;
      if (tt < nt) {
        char* nk = smem + buf * STAGE + j * 2 * C::TILE;
        stage64<D, RAGGED, NW, (D >= 128)>(kbase + (size_t)tt * 64 * tstride, tstride, nk, wave, lane, L - tt * 64);
        stage64<D, RAGGED, NW, (D >= 128)>(vbase + (size_t)tt * 64 * tstride, tstride, nk + C::TILE, wave, lane, L - tt * 64);
        if (DROP) stage_keyhash(skh[buf] + j * 16, seed, tt, tid);
      }
    }
  };
  stage_super(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef O2_STAMP
  unsigned tS = 0, tQK = 0, tSM = 0, tPV = 0, tW = 0, tB = 0, t0_ = O2_T(), t1_;
  const unsigned tstart_ = t0_;
#endif
  // scores of tile t relative to `ref` (exp2 domain): S^T[kb] = K_kb . Q~^T - ref  (rows = keys in registers, column =
  // query on the lane).  The two key blocks' chains are interleaved: a K fragment is consumed two MFMAs after the
  // previous one of its chain, so its LDS read has twice the time to land.
  // `init` (16 registers, all = -reference) is the C operand of both chains' first MFMA and stays live: no per-tile
  // accumulator initialisation.
  auto scores = [&](const char* sk, int t, const f32x16& init, f32x16 (&s)[2]) {
    if constexpr (D == 256) {             // one wave per SIMD: fragments in flight ahead of their MFMAs, pinned (see attn_fwd_kernel)
      constexpr int LA = O2_FWD256_LA;
      bf16x8 k0[LA + 1], k1[LA + 1];
#pragma unroll
      for (int i = 0; i < LA; ++i) {
        k0[i] = row_frag<D>(sk, (lane & 31), i, hq);
        k1[i] = row_frag<D>(sk, 32 + (lane & 31), i, hq);
      }
#pragma unroll
      for (int ds = 0; ds < C::NDS; ++ds) {
        if (ds + LA < C::NDS) {
          k0[(ds + LA) % (LA + 1)] = row_frag<D>(sk, (lane & 31), ds + LA, hq);
          k1[(ds + LA) % (LA + 1)] = row_frag<D>(sk, 32 + (lane & 31), ds + LA, hq);
        }
        s[0] = MFMA32(k0[ds % (LA + 1)], qf[ds], ds == 0 ? init : s[0]);
        s[1] = MFMA32(k1[ds % (LA + 1)], qf[ds], ds == 0 ? init : s[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int ds = 0; ds < C::NDS; ++ds)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          s[kb] = MFMA32(row_frag<D>(sk, kb * 32 + (lane & 31), ds, hq), qf[ds], ds == 0 ? init : s[kb]);
    }
    if (RAGGED && t == nt - 1 && (L & 63)) {   // keys past the end of a ragged sequence get no weight
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hq >= L) s[kb][r] = -1e30f;
    }
  };
  auto rowmax = [&](const f32x16 (&s)[2]) {
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    return fmaxf(mx, __shfl_xor(mx, 32));
  };
  {   // reference maximum of every row = its maximum over the first tile (one extra S product per workgroup row block)
    f32x16 s0[2], zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    scores(smem, 0, zero, s0);
    m_run = rowmax(s0);
  }
  // first tile of a super-tile: put the next super-tile in flight
  auto stage_next = [&](int t) {
    if (t % KT == 0 && (t / KT + 1) * KT < nt) stage_super(t / KT + 1, cur ^ 1);
  };
  // O^T[db] += V^T . (dropout(P))^T, then (last tile of a super-tile) the hand-over: next super-tile landed, everybody done
  auto drop_pv_sync = [&](int t, const char* sv, f32x16 (&s)[2]) {
    if (DROP) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        drop_keys_in_regs<true>(s[kb], rowhash, *reinterpret_cast<const u32x4*>(&skh[cur][(t % KT) * 16 + hq * 8 + kb * 4]), thr);
    }
    O2_SEG(tSM)
    if constexpr (D == 256) {
      bf16x8 pf[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) pf[g] = pack_frag(s[g >> 1], g & 1);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int NV = 4 * C::NDB, LA = O2_FWD256_LA;
      bf16x8 vr[LA + 1];
#pragma unroll
      for (int i = 0; i < LA; ++i) vr[i] = tr_frag<D>(sv, (i / C::NDB) * 16, i % C::NDB, lane);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (i + LA < NV) vr[(i + LA) % (LA + 1)] = tr_frag<D>(sv, ((i + LA) / C::NDB) * 16, (i + LA) % C::NDB, lane);
        o[i % C::NDB] = MFMA32(vr[i % (LA + 1)], pf[i / C::NDB], o[i % C::NDB]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 pf = pack_frag(s[kb], ss);
#pragma unroll
          for (int db = 0; db < C::NDB; ++db)
            o[db] = MFMA32(tr_frag<D>(sv, kb * 32 + ss * 16, db, lane), pf, o[db]);
        }
    }
    O2_SEG(tPV)
    if (t % KT == KT - 1 || t == nt - 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      O2_SEG(tW)
      __syncthreads();
      O2_SEG(tB)
      cur ^= 1;
    }
  };
  // ---- fast loop: FIXED reference (the row's maximum over the first tile).  P, O and l share that one reference per row and
  // fp32 / bf16 keep their relative precision at any scale, so as long as no probability relative to it overflows the result
  // is exact -- and a score element costs one exp2, one add and its share of the bf16 conversion: no multiply, no
  // subtraction, no row maximum, no rescale of O.  The row sums the loop computes anyway are the guard: a wave in which some
  // half-row sum of a tile leaves [0, 2^40] (inf and NaN included) drops, from that tile on, to the safe loop below.
  constexpr float FAST_LIMIT = 1.0995116e12f;      // 2^40
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = -m_run;
  int t = 0;
  bool fast_ok = true;
  for (; t < nt; ++t) {
    const char* sk = smem + cur * STAGE + (t % KT) * 2 * C::TILE;
    stage_next(t);
    O2_SEG(tS)
    f32x16 s[2];
    scores(sk, t, negm, s);
    O2_SEG(tQK)
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[kb][r]);
        s[kb][r] = p;
        psum += p;
      }
    if (__any(!(psum <= FAST_LIMIT))) { fast_ok = false; break; }     // wave-uniform; tile t is redone by the safe loop
    l_run += psum;
    drop_pv_sync(t, sk + C::TILE, s);
  }
  // ---- safe loop (only after the guard tripped; tile t + 1 is already in flight): online softmax with a moving reference,
  // rescaled when some row of the wave outgrows it by more than 2^RESCALE_THR (wave-uniform, deferred: P is then bounded by
  // 2^RESCALE_THR instead of 1; O and l use the same reference, the result is exact).
  if (!fast_ok) {
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    bool staged = true;
    for (; t < nt; ++t) {
      const char* sk = smem + cur * STAGE + (t % KT) * 2 * C::TILE;
      if (!staged) stage_next(t);
      staged = false;
      f32x16 s[2];
      scores(sk, t, zero, s);
      const float mt = rowmax(s);
      constexpr float RESCALE_THR = 5.0f;
      float alpha = 1.0f;
      if (__any(mt > m_run + RESCALE_THR)) {
        const float m_new = fmaxf(m_run, mt);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < C::NDB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[kb][r] - m_run);
          s[kb][r] = p;
          psum += p;
        }
      l_run = l_run * alpha + psum;
      drop_pv_sync(t, sk + C::TILE, s);
    }
  }
#ifdef O2_STAMP
  if (blockIdx.x < 64 && wave == 0 && lane == 0) {
    unsigned* dd_ = o2_dbg_attn + blockIdx.x * 8;
    dd_[0] = tS; dd_[1] = tQK; dd_[2] = tSM; dd_[3] = tPV; dd_[4] = tW; dd_[5] = tB; dd_[6] = O2_T() - tstart_; dd_[7] = (unsigned)nt;
  }
#endif
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = (DROP ? dscale : 1.0f) / l_tot;   // dropout scale folded out of the inner loop
  if (!q_ok) return;
  if (hq == 0) lse[((size_t)(b * H + head)) * L + qrow] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
  bf16_t* orow = out + ((size_t)b * L + qrow) * (size_t)ldo + (size_t)head * D;   // ldo: token-row pitch of out (>= H * D)
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dd = db * 32 + 8 * g4 + 4 * hq;
      u32x2 w;
      w[0] = pack_bf2(o[db][4 * g4] * inv, o[db][4 * g4 + 1] * inv);
      w[1] = pack_bf2(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv);
      *reinterpret_cast<u32x2*>(orow + dd) = w;
    }
}

// =============================================================================================
// forward, d = 128, q stored pre-scaled, L % 256 == 0: the GENERATED kernel (tools/gen_attn_fwd.py -> attn_fwd_asm.h)
// =============================================================================================
// One wave per SIMD with the whole register file; a wave owns 64 query rows as two 32-row blocks, the vector work of one block's
// tile (exp2, row sums, dropout mask, bf16 packing) sits in the MFMA gaps of the other block's S / O products, every K / V
// fragment read from LDS feeds two MFMAs, K / V tiles arrive by LDS-DMA one tile ahead of their first read, the softmax keeps a
// fixed reference per row (guard + out-of-line fix-up).  The statement below is the whole kernel body: every instruction of it
// is placed by the generator (its header describes the schedule); tools/cdna_emu.py executes the same text on the CPU
// (tests/test_attn_asm_emu_cpu.py).  What the compiler contributes: the workgroup's coordinates, the key-group hash table of the
// sequence (LDS, read by the dropout mask) and the lane's two row hashes.
// Contract audited by tests/test_w4_audit_cpu.py: no scratch, no spills, the statement's registers (v8-v255, a0-a255,
// s36-s69) are clobbers so the descriptor allocates 512 registers per lane.
#define O2_AF_MAX_L 16384
// The generated streams address K / V tiles, query rows and output rows with 32-bit BYTE offsets from per-(sample, head) base
// pointers (strideb = pitch * 2; offsets up to L * pitch * 2): the launchers take the generated kernels only when those offsets
// stay below 2^31 and fall back to the compiler-scheduled kernels (64-bit addressing) otherwise.
static inline bool attn_w4_range_ok(int L, int ldq, int ldo) {
  const uint64_t lim = 1ull << 31;
  return (uint64_t)(L + 64) * (uint64_t)ldq * 2ull < lim && (uint64_t)(L + 64) * (uint64_t)ldo * 2ull < lim;
}
template <bool DROP>
__global__ __launch_bounds__(256, 1) void attn_fwd_w4_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int L, int H, unsigned thr, float dscale,
                                                            uint64_t seed_arg, int ldo, int ldq) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(1024))) char smem[O2_AF_LDS_BYTES(O2_AF_MAX_L)];   // [2 slots][K 16 KiB | V 16 KiB] | key-group hashes
  const uint64_t seed = seed_arg ^ o2_seed_salt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_i, head, b;
  attn_tile_coords(L / 256, H, tile_i, head, b);
  const int q0 = tile_i * 256 + wave * 64;
  const size_t tstride = (size_t)ldq;      // token-row pitch of qkv (and dqkv), >= 3 * H * D
  if (DROP) {
    // key-group hashes of the whole sequence (they depend on neither batch nor head): position T*16 + h*8 + j holds
    // K(T*16 + 2 j + h), so lane half h reads the 8 values of its keys of tile T as two 16-byte pieces
    uint32_t* skh = reinterpret_cast<uint32_t*>(smem + O2_AF_KH_OFF);
    for (int i = tid; i < L / 4 + 16; i += 256) {
      const uint32_t T = (uint32_t)i >> 4, wq = (uint32_t)i & 15;
      skh[i] = o2_attn_keyhash(seed, T * 16 + 2 * (wq & 7) + (wq >> 3));
    }
  }
  __syncthreads();
  const char* kptr = reinterpret_cast<const char*>(qkv + (size_t)b * L * tstride + (size_t)H * D + (size_t)head * D);
  const char* qptr = reinterpret_cast<const char*>(qkv + ((size_t)b * L + q0) * tstride + (size_t)head * D);
  char* optr = reinterpret_cast<char*>(out + ((size_t)b * L + q0) * (size_t)ldo + (size_t)head * D);
  char* lptr = reinterpret_cast<char*>(lse + ((size_t)(b * H + head)) * L + q0);
  const uint64_t row = (uint64_t)(b * H + head) * L + (uint64_t)(q0 + (lane & 31));
  const uint32_t rhx = DROP ? o2_attn_rowhash(seed, row) : 0u, rhy = DROP ? o2_attn_rowhash(seed, row + 32) : 0u;
  const uint32_t ldsb = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const int nt = L / 64, strideb = (int)(tstride * 2), hd2 = H * D * 2;
  const uint32_t dsc = __float_as_uint(dscale);
  if constexpr (DROP) {
    asm volatile(O2_AF_ASM_DROP
                 :
                 : [kptr] "s"(kptr), [qptr] "s"(qptr), [optr] "s"(optr), [lptr] "s"(lptr), [nt] "s"(nt), [strideb] "s"(strideb),
                   [hd2] "s"(hd2), [ldsb] "s"(ldsb), [wave] "s"(wave), [thr] "s"(thr), [dscale] "s"(dsc), [orowb] "s"(ldo * 2),
                   [rhx] "v"(rhx), [rhy] "v"(rhy)
                 : O2_AF_CLOBBERS);
  } else {
    asm volatile(O2_AF_ASM_NODROP
                 :
                 : [kptr] "s"(kptr), [qptr] "s"(qptr), [optr] "s"(optr), [lptr] "s"(lptr), [nt] "s"(nt), [strideb] "s"(strideb),
                   [hd2] "s"(hd2), [ldsb] "s"(ldsb), [wave] "s"(wave), [thr] "s"(thr), [dscale] "s"(dsc), [orowb] "s"(ldo * 2),
                   [rhx] "v"(rhx), [rhy] "v"(rhy)
                 : O2_AF_CLOBBERS);
  }
}

// =============================================================================================
// per-row statistics of the backward, ready to use: ws[0][b,h,q] = -lse * log2(e), ws[1][b,h,q] = -(sum_d dO*O) / dscale
// =============================================================================================
// Rows are padded per (b, h) to Lp >= ceil64(L) + 64 entries, the pad holding (-1e30, 0): the dK / dV kernels copy a tile's 64
// (or 32) entries straight into LDS by LDS-DMA -- no clamping, no conversion, no register -- and a query row past the end of the
// sequence contributes exp2(s - 1e30) = 0.
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         const float* __restrict__ lse, float* __restrict__ ws, int B, int L,
                                                         int H, int D, int Lp, float inv_dscale, int ldo) {
  // one 16-lane group per (batch, head, padded token) row of D elements, the TOKEN fastest: a workgroup's 16 groups write 16
  // consecutive floats of each statistics row (64-byte pieces instead of sixteen 4-byte writes Lp floats apart) and read sixteen
  // 2 D-byte row pieces one token pitch apart
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int li = threadIdx.x & 15;
  const int64_t nrows = (int64_t)B * Lp * H;
  if (grp >= nrows) return;
  const int64_t bh = grp / Lp;
  const int64_t q = grp - bh * Lp;
  const int64_t bb = bh / H;
  const int hh = (int)(bh - bb * H);
  float* nlse2 = ws + ((size_t)(bb * H + hh)) * Lp + q;
  float* ndelta = nlse2 + (size_t)B * H * Lp;
  if (q >= L) {                                   // pad row
    if (li == 0) { *nlse2 = -1e30f; *ndelta = 0.f; }
    return;
  }
  const int64_t row = (bb * L + q) * H + hh;
  const bf16_t* po = o + (bb * L + q) * (int64_t)ldo + (int64_t)hh * D;   // out may carry a token-row pitch; dout is contiguous
  const bf16_t* pd = dout + row * D;
  float s = 0.f;
  for (int c = li; c < D / 8; c += 16) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(po + c * 8);
    const u32x4 d = *reinterpret_cast<const u32x4*>(pd + c * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      s += bf2f((bf16_t)(a[j] & 0xffff)) * bf2f((bf16_t)(d[j] & 0xffff)) +
           bf2f((bf16_t)(a[j] >> 16)) * bf2f((bf16_t)(d[j] >> 16));
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (li == 0) {
    *nlse2 = -(lse[((size_t)(bb * H + hh)) * L + q] * 1.4426950408889634f);
    *ndelta = -(s * inv_dscale);
  }
}

// =============================================================================================
// backward, dQ: same geometry as the forward (query on the lane)
// =============================================================================================
template <int D, bool DROP, bool RAGGED, int NW>
__global__ __launch_bounds__(NW * 64, (D == 256 ? 1 : 2)) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv,
                                                             const bf16_t* __restrict__ dout,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ delta,
                                                             bf16_t* __restrict__ dqkv, int L, int H, float scale,
                                                             unsigned thr, float dscale, uint64_t seed_arg, float opmul, int Lp, int ldq) {
  const uint64_t seed = seed_arg ^ o2_seed_salt;   // see common.h: fresh masks for every replay of a captured step
  using C = Cfg<D>;
  __shared__ __attribute__((aligned(16))) char smem[4 * C::TILE];
  __shared__ __attribute__((aligned(16))) uint32_t skh[2][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hq = lane >> 5;
  int tile_i, head, b;
  attn_tile_coords((L + NW * 32 - 1) / (NW * 32), H, tile_i, head, b);
  const int q0 = tile_i * (NW * 32) + wave * 32;
  const size_t tstride = (size_t)ldq;      // token-row pitch of qkv (and dqkv), >= 3 * H * D
  const bf16_t* qbase = qkv + (size_t)b * L * tstride + (size_t)head * D;
  const bf16_t* kbase = qbase + (size_t)H * D;
  const bf16_t* vbase = qbase + (size_t)2 * H * D;
  const int qrow_raw = q0 + (lane & 31);
  const bool q_ok = !RAGGED || qrow_raw < L;
  const int qrow = q_ok ? qrow_raw : L - 1;
  const float sc_log2 = opmul;     // multiplier of the register-resident S operand: log2(e)/sqrt(d), or 1 when q is stored pre-scaled

  bf16x8 qf[C::NDS], dof[C::NDS];
  const bf16_t* dorow = dout + ((size_t)b * L + qrow) * ((size_t)H * D) + (size_t)head * D;
#pragma unroll
  for (int ds = 0; ds < C::NDS; ++ds) {
    qf[ds] = scale_frag(*reinterpret_cast<const bf16x8*>(qbase + (size_t)qrow * tstride + ds * 16 + 8 * hq), sc_log2);
    dof[ds] = *reinterpret_cast<const bf16x8*>(dorow + ds * 16 + 8 * hq);
  }
  // Q enters the S MFMAs pre-multiplied by log2(e)/sqrt(d) and the S / dP accumulators START at -lse2 / -delta of the lane's
  // row (two 16-register constant blocks, the C operands of every chain's first MFMA): S leaves the matrix pipe as the exp2
  // argument and dP as dP - delta -- no multiply and no subtraction per score element.
  // (lse / delta: the ready-made tables of attn_delta_kernel, -lse log2(e) and -delta / dscale, padded row stride Lp)
  const size_t sidx_p = ((size_t)(b * H + head)) * Lp + qrow;
  const float lse2 = -lse[sidx_p];
  const float dlt = -delta[sidx_p];
  // (d = 128 runs two waves per SIMD on 256 registers: only the S block fits there, dP - delta stays a subtraction)
  constexpr bool DLT_INIT = (D != 128);
  f32x16 nlse, ndlt;
#pragma unroll
  for (int r = 0; r < 16; ++r) { nlse[r] = -lse2; ndlt[r] = DLT_INIT ? -dlt : 0.f; }
  const uint32_t rowhash = DROP ? o2_attn_rowhash(seed, (uint64_t)(b * H + head) * L + (uint64_t)qrow) : 0u;

  f32x16 dq[C::NDB];
#pragma unroll
  for (int i = 0; i < C::NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

  const int nt = (L + 63) / 64;
  stage64<D, RAGGED, NW, DROP>(kbase, tstride, smem, wave, lane, L);
  stage64<D, RAGGED, NW, DROP>(vbase, tstride, smem + C::TILE, wave, lane, L);
  if (DROP) stage_keyhash(skh[0], seed, 0, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sk = smem + cur * 2 * C::TILE;
    const char* sv = sk + C::TILE;
    if (t + 1 < nt) {
      char* nk = smem + (cur ^ 1) * 2 * C::TILE;
      stage64<D, RAGGED, NW, DROP>(kbase + (size_t)(t + 1) * 64 * tstride, tstride, nk, wave, lane, L - (t + 1) * 64);
      stage64<D, RAGGED, NW, DROP>(vbase + (size_t)(t + 1) * 64 * tstride, tstride, nk + C::TILE, wave, lane, L - (t + 1) * 64);
      if (DROP) stage_keyhash(skh[cur ^ 1], seed, t + 1, tid);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
      if constexpr (D == 256) {       // one wave per SIMD: O2_FWD256_LA k-steps of K / V fragments in flight, pinned (see attn_fwd_kernel)
        constexpr int LA = O2_FWD256_LA;
        bf16x8 kr[LA + 1], vr[LA + 1];
#pragma unroll
        for (int i = 0; i < LA; ++i) {
          kr[i] = row_frag<D>(sk, kb * 32 + (lane & 31), i, hq);
          vr[i] = row_frag<D>(sv, kb * 32 + (lane & 31), i, hq);
        }
#pragma unroll
        for (int ds = 0; ds < C::NDS; ++ds) {
          if (ds + LA < C::NDS) {
            kr[(ds + LA) % (LA + 1)] = row_frag<D>(sk, kb * 32 + (lane & 31), ds + LA, hq);
            vr[(ds + LA) % (LA + 1)] = row_frag<D>(sv, kb * 32 + (lane & 31), ds + LA, hq);
          }
          s = MFMA32(kr[ds % (LA + 1)], qf[ds], ds == 0 ? nlse : s);
          dp = MFMA32(vr[ds % (LA + 1)], dof[ds], ds == 0 ? ndlt : dp);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int ds = 0; ds < C::NDS; ++ds) {
          s = MFMA32(row_frag<D>(sk, kb * 32 + (lane & 31), ds, hq), qf[ds], ds == 0 ? nlse : s);
          dp = MFMA32(row_frag<D>(sv, kb * 32 + (lane & 31), ds, hq), dof[ds], ds == 0 ? ndlt : dp);
        }
      }
      // a dropped element: (0 - delta) instead of (dP - delta)
      if (DROP)
        drop_keys_in_regs<false>(dp, rowhash, *reinterpret_cast<const u32x4*>(&skh[cur][hq * 8 + kb * 4]), thr,
                                 DLT_INIT ? -dlt : 0.f);
      const bool tail = RAGGED && (t == nt - 1) && (L & 63);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(s[r]);
        if (tail && t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hq >= L) p = 0.f;   // key past the end
        s[r] = DLT_INIT ? p * dp[r] : p * (dp[r] - dlt);  // dS^T
      }
      if constexpr (D == 256) {
        bf16x8 dsf[2];
        dsf[0] = pack_frag(s, 0);
        dsf[1] = pack_frag(s, 1);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NV = 2 * C::NDB, LA = O2_FWD256_LA;
        bf16x8 tr[LA + 1];
#pragma unroll
        for (int i = 0; i < LA; ++i) tr[i] = tr_frag<D>(sk, kb * 32 + (i / C::NDB) * 16, i % C::NDB, lane);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          if (i + LA < NV) tr[(i + LA) % (LA + 1)] = tr_frag<D>(sk, kb * 32 + ((i + LA) / C::NDB) * 16, (i + LA) % C::NDB, lane);
          dq[i % C::NDB] = MFMA32(tr[i % (LA + 1)], dsf[i / C::NDB], dq[i % C::NDB]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 dsf = pack_frag(s, ss);
#pragma unroll
          for (int db = 0; db < C::NDB; ++db)
            dq[db] = MFMA32(tr_frag<D>(sk, kb * 32 + ss * 16, db, lane), dsf, dq[db]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  if (!q_ok) return;
  bf16_t* drow = dqkv + ((size_t)b * L + qrow) * tstride + (size_t)head * D;
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dd = db * 32 + 8 * g4 + 4 * hq;
      u32x2 w;
      const float fs = DROP ? scale * dscale : scale;
      w[0] = pack_bf2(dq[db][4 * g4] * fs, dq[db][4 * g4 + 1] * fs);
      w[1] = pack_bf2(dq[db][4 * g4 + 2] * fs, dq[db][4 * g4 + 3] * fs);
      *reinterpret_cast<u32x2*>(drow + dd) = w;
    }
}

// =============================================================================================
// backward, dQ, d = 128, q stored pre-scaled, L % 256 == 0: the GENERATED kernel (tools/gen_attn_dq.py -> attn_dq_asm.h)
// =============================================================================================
// Same construction as attn_fwd_w4_kernel (one wave per SIMD, two 32-row blocks per wave, every instruction placed by the
// generator, the same text executed on the CPU by tests/test_attn_dq_asm_emu_cpu.py); the schedule is described in the
// generator's header.  The compiler contributes the coordinates, the key-group hash table and the lane's two row hashes.
template <bool DROP>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_w4_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ nlse2, const float* __restrict__ ndelta,
                                                               bf16_t* __restrict__ dqkv, int L, int H, unsigned thr, float fs,
                                                               uint64_t seed_arg, int Lp, int ldq) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(1024))) char smem[O2_DQ_LDS_BYTES(O2_AF_MAX_L)];   // 4 x [K 16 KiB] | 4 x [V 16 KiB] | key-group hashes
  const uint64_t seed = seed_arg ^ o2_seed_salt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_i, head, b;
  attn_tile_coords(L / 256, H, tile_i, head, b);
  const int q0 = tile_i * 256 + wave * 64;
  const size_t tstride = (size_t)ldq;      // token-row pitch of qkv (and dqkv), >= 3 * H * D
  if (DROP) {
    uint32_t* skh = reinterpret_cast<uint32_t*>(smem + O2_DQ_KH_OFF);
    for (int i = tid; i < L / 4 + 16; i += 256) {
      const uint32_t T = (uint32_t)i >> 4, wq = (uint32_t)i & 15;
      skh[i] = o2_attn_keyhash(seed, T * 16 + 2 * (wq & 7) + (wq >> 3));
    }
  }
  __syncthreads();
  const char* kptr = reinterpret_cast<const char*>(qkv + (size_t)b * L * tstride + (size_t)H * D + (size_t)head * D);
  const char* qptr = reinterpret_cast<const char*>(qkv + ((size_t)b * L + q0) * tstride + (size_t)head * D);
  const char* doptr = reinterpret_cast<const char*>(dout + (((size_t)b * L + q0) * H + head) * D);
  char* optr = reinterpret_cast<char*>(dqkv + ((size_t)b * L + q0) * tstride + (size_t)head * D);
  const char* lseptr = reinterpret_cast<const char*>(nlse2 + ((size_t)(b * H + head)) * Lp + q0);
  const char* dltptr = reinterpret_cast<const char*>(ndelta + ((size_t)(b * H + head)) * Lp + q0);
  const uint64_t row = (uint64_t)(b * H + head) * L + (uint64_t)(q0 + (lane & 31));
  const uint32_t rhx = DROP ? o2_attn_rowhash(seed, row) : 0u, rhy = DROP ? o2_attn_rowhash(seed, row + 32) : 0u;
  const uint32_t ldsb = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const int nt = L / 64, strideb = (int)(tstride * 2), hd2 = H * D * 2;
  const uint32_t fsb = __float_as_uint(fs);
#define O2_DQ_OPERANDS                                                                                                          \
  [kptr] "s"(kptr), [qptr] "s"(qptr), [optr] "s"(optr), [doptr] "s"(doptr), [lseptr] "s"(lseptr), [dltptr] "s"(dltptr),        \
      [nt] "s"(nt), [strideb] "s"(strideb), [hd2] "s"(hd2), [ldsb] "s"(ldsb), [wave] "s"(wave), [thr] "s"(thr), [fs] "s"(fsb), \
      [dorowb] "s"(hd2), [rhx] "v"(rhx), [rhy] "v"(rhy)
  if constexpr (DROP) {
    asm volatile(O2_DQ_ASM_DROP : : O2_DQ_OPERANDS : O2_DQ_CLOBBERS);
  } else {
    asm volatile(O2_DQ_ASM_NODROP : : O2_DQ_OPERANDS : O2_DQ_CLOBBERS);
  }
#undef O2_DQ_OPERANDS
}

// =============================================================================================
// backward, dK/dV: key on the lane; the workgroup owns NW*32 keys (32 per wave) and sweeps all queries
// =============================================================================================
// WHICH: 0 = dK and dV in one pass (d = 64: fits 2 waves/SIMD);  1 = dK only;  2 = dV only.
// At d = 128 the fused form needs 236 VGPR + 160 AGPR (1 wave/SIMD, measured 576 TFLOP/s executed); split in two
// passes (5 MFMA products instead of 4) each pass fits 2 waves/SIMD.
template <int D, bool DROP, int WHICH, bool RAGGED, int NW>
__global__ __launch_bounds__(NW * 64, (D == 256 ? 1 : 2)) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv,
                                                              const bf16_t* __restrict__ dout,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta,
                                                              bf16_t* __restrict__ dqkv, int L, int H, float scale,
                                                              unsigned thr, float dscale, uint64_t seed_arg, float opmul,
                                                              float kgrad, int Lp, int ldq) {
  const uint64_t seed = seed_arg ^ o2_seed_salt;   // see common.h: fresh masks for every replay of a captured step
  using C = Cfg<D>;
  constexpr bool DO_DK = WHICH != 2, DO_DV = WHICH != 1;
  __shared__ __attribute__((aligned(16))) char smem[4 * C::TILE + 2 * 3 * 64 * 4];  // [2][Q|dO] + [2][lse2|delta|row hash]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hq = lane >> 5;
  int tile_i, head, b;
  attn_tile_coords((L + NW * 32 - 1) / (NW * 32), H, tile_i, head, b);
  const int k0 = tile_i * (NW * 32) + wave * 32;
  const size_t tstride = (size_t)ldq;      // token-row pitch of qkv (and dqkv), >= 3 * H * D
  const size_t ostride = (size_t)H * D;
  const bf16_t* qbase = qkv + (size_t)b * L * tstride + (size_t)head * D;
  const bf16_t* kbase = qbase + (size_t)H * D;
  const bf16_t* vbase = qbase + (size_t)2 * H * D;
  const bf16_t* dobase = dout + (size_t)b * L * ostride + (size_t)head * D;
  const int krow_raw = k0 + (lane & 31);
  const bool k_ok = !RAGGED || krow_raw < L;
  const int krow = k_ok ? krow_raw : L - 1;
  const float sc_log2 = opmul;     // multiplier of the register-resident S operand: log2(e)/sqrt(d), or 1 when q is stored pre-scaled
  float* sstat = reinterpret_cast<float*>(smem + 4 * C::TILE);

  bf16x8 kf[C::NDS], vf[DO_DK ? C::NDS : 1];
#pragma unroll
  for (int ds = 0; ds < C::NDS; ++ds) {
    kf[ds] = scale_frag(*reinterpret_cast<const bf16x8*>(kbase + (size_t)krow * tstride + ds * 16 + 8 * hq), sc_log2);
    if (DO_DK) vf[ds] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)krow * tstride + ds * 16 + 8 * hq);
  }
  f32x16 dk[DO_DK ? C::NDB : 1], dv[DO_DV ? C::NDB : 1];
#pragma unroll
  for (int i = 0; i < C::NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (DO_DK) dk[i][r] = 0.f;
      if (DO_DV) dv[i][r] = 0.f;
    }

  const uint64_t bh = (uint64_t)(b * H + head);
  const uint32_t keyhash = DROP ? o2_attn_keyhash(seed, (uint32_t)(krow >> 2)) : 0u;   // this lane's key group
  const int nt = (L + 63) / 64;
  // per-row statistics of a tile (-lse log2(e), -delta / dscale: the ready-made, padded tables of attn_delta_kernel) go
  // global -> LDS by two 256-byte LDS-DMA pieces, in the same vmcnt stream as the tile's own pieces: no register, no conversion
  // and -- unlike the load + convert + ds_write of rounds 1-2, whose first use of the loaded word sat at the top of the loop body
  // -- no wait for the load's latency and for every DMA piece issued before it, once per tile (round 3: -24 % on the d = 256
  // dK+dV kernel, profiles/r03_attn_dkv_stats_ab.txt).  The accumulators of S and dP start from these rows.
  const size_t sbase_p = ((size_t)(b * H + head)) * Lp;
  auto stage_stats = [&](int t, int buf) {
    if (wave < 2) {
      int ln = lane;
      asm volatile("" : "+v"(ln));        // the per-lane offset is rebuilt here, not kept in a register across the tile loop
      glds4_asm((wave ? delta : lse) + sbase_p + (size_t)t * 64, (uint32_t)ln * 4u,
                (uint32_t)(uintptr_t)LDS_PTR(float, sstat + (buf * 3 + wave) * 64));
    } else if (DROP && tid < 192) {   // dropout: hashes of the tile's 64 query rows
      int i = tid & 63;
      asm volatile("" : "+v"(i));         // (rebuilt here: hoisted out of the tile loop the slot address gets spilled, and its
                                          //  reload's vmcnt(0) drains the DMA pieces just issued)
      reinterpret_cast<uint32_t*>(sstat)[(buf * 3 + 2) * 64 + i] = o2_attn_rowhash(seed, bh * (uint64_t)L + (uint64_t)(t * 64 + i));
    }
  };
  stage64<D, RAGGED, NW, DROP>(qbase, tstride, smem, wave, lane, L);
  stage64<D, RAGGED, NW, DROP>(dobase, ostride, smem + C::TILE, wave, lane, L);
  stage_stats(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sq = smem + cur * 2 * C::TILE;
    const char* sdo = sq + C::TILE;
    if (t + 1 < nt) {
      char* nq = smem + (cur ^ 1) * 2 * C::TILE;
      stage64<D, RAGGED, NW, DROP>(qbase + (size_t)(t + 1) * 64 * tstride, tstride, nq, wave, lane, L - (t + 1) * 64);
      stage64<D, RAGGED, NW, DROP>(dobase + (size_t)(t + 1) * 64 * ostride, ostride, nq + C::TILE, wave, lane, L - (t + 1) * 64);
      stage_stats(t + 1, cur ^ 1);
    }
    const float* s_lse = sstat + (cur * 3 + 0) * 64;
    const float* s_dlt = sstat + (cur * 3 + 1) * 64;
    const uint32_t* s_rh = reinterpret_cast<const uint32_t*>(sstat) + (cur * 3 + 2) * 64;
    // the dV-only pass has a single chain per query block (S = Q.K^T): run both blocks' chains interleaved so each
    // Q fragment's LDS read gets two MFMAs of time to land (the dK pass interleaves its S and dP chains instead)
    // (K is pre-multiplied by log2(e)/sqrt(d) and the accumulators START at -lse2[q] / -delta[q], read from the tile's LDS
    //  table: S leaves the matrix pipe as the exp2 argument, dP as dP - delta)
    auto init_rows = [&](const float* tab, int qb, f32x16& x) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(tab + qb * 32 + 8 * g4 + 4 * hq);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[4 * g4 + e] = v4[e];
      }
    };
    f32x16 s_pre[(!DO_DK) ? 2 : 1];
    if (!DO_DK) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) init_rows(s_lse, qb, s_pre[qb]);
#pragma unroll
      for (int ds = 0; ds < C::NDS; ++ds)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          s_pre[qb] = MFMA32(row_frag<D>(sq, qb * 32 + (lane & 31), ds, hq), kf[ds], s_pre[qb]);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      // S[q x key] = Q . K^T, dP[q x key] = dO . V^T  (rows = queries in registers, column = key on the lane)
      f32x16 s, dp;
      if (DO_DK) {
        init_rows(s_lse, qb, s);
        init_rows(s_dlt, qb, dp);
#pragma unroll
        for (int ds = 0; ds < C::NDS; ++ds) {
          s = MFMA32(row_frag<D>(sq, qb * 32 + (lane & 31), ds, hq), kf[ds], s);
          dp = MFMA32(row_frag<D>(sdo, qb * 32 + (lane & 31), ds, hq), vf[ds], dp);
        }
      } else {
        s = s_pre[qb];
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
      }
      // dropout: one mask word covers 4 consecutive keys = the 4 lanes of a quad; each lane mixes 4 of the 16
      // query rows (rows r with (r&3) == lane&3: their hashes come from the tile's LDS table) with its key-group
      // hash and the quad shares the words by DPP broadcast.
      const int kbyte = 8 * (krow & 3);
      // per-row statistics: registers 4g..4g+3 are 4 consecutive query rows -> one 16-byte LDS read per group,
      // fetched just in time (keeps 32 VGPRs out of the MFMA section's live set)
      uint32_t hmine[4] = {0u, 0u, 0u, 0u};
      if (DROP) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          hmine[g4] = o2_attn_mix(s_rh[qb * 32 + 8 * g4 + 4 * hq + (lane & 3)], keyhash);
      }
      f32x16 pd;  // P after dropout (for dV)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 ndl4 = {0.f, 0.f, 0.f, 0.f};
        if (DO_DK && DROP) ndl4 = *reinterpret_cast<const f32x4*>(s_dlt + qb * 32 + 8 * g4 + 4 * hq);   // -delta
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g4 + e;
          float p = __builtin_amdgcn_exp2f(s[r]);
          float dpr = dp[r];            // dP - delta
          float pdr = p;
          if (DROP) {
            const uint32_t hh = quad_bcast(hmine[g4], e);
            const bool keep = ((hh >> kbyte) & 0xffu) >= thr;
            dpr = keep ? dpr : ndl4[e];     // a dropped element: 0 - delta  (dscale is applied once, on the final dK / dV tiles)
            pdr = keep ? p : 0.f;
          }
          if (DO_DV) pd[r] = pdr;
          if (DO_DK) s[r] = p * dpr;  // dS
        }
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        if (DO_DV) {
          const bf16x8 pf = pack_frag(pd, ss);
#pragma unroll
          for (int db = 0; db < C::NDB; ++db)
            dv[db] = MFMA32(tr_frag<D>(sdo, qb * 32 + ss * 16, db, lane), pf, dv[db]);   // dV^T += dO^T . P
        }
        if (DO_DK) {
          const bf16x8 dsf = pack_frag(s, ss);
#pragma unroll
          for (int db = 0; db < C::NDB; ++db)
            dk[db] = MFMA32(tr_frag<D>(sq, qb * 32 + ss * 16, db, lane), dsf, dk[db]);   // dK^T += Q^T . dS
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  if (!k_ok) return;
  bf16_t* dkrow = dqkv + ((size_t)b * L + krow) * tstride + (size_t)H * D + (size_t)head * D;
  bf16_t* dvrow = dkrow + (size_t)H * D;
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dd = db * 32 + 8 * g4 + 4 * hq;
      u32x2 w;
      if (DO_DK) {
        const float fk = DROP ? kgrad * dscale : kgrad;    // 1/sqrt(d), or ln 2 when the Q rows in LDS are pre-scaled
        w[0] = pack_bf2(dk[db][4 * g4] * fk, dk[db][4 * g4 + 1] * fk);
        w[1] = pack_bf2(dk[db][4 * g4 + 2] * fk, dk[db][4 * g4 + 3] * fk);
        *reinterpret_cast<u32x2*>(dkrow + dd) = w;
      }
      if (DO_DV) {
        const float fv = DROP ? dscale : 1.0f;
        w[0] = pack_bf2(dv[db][4 * g4] * fv, dv[db][4 * g4 + 1] * fv);
        w[1] = pack_bf2(dv[db][4 * g4 + 2] * fv, dv[db][4 * g4 + 3] * fv);
        *reinterpret_cast<u32x2*>(dvrow + dd) = w;
      }
    }
}

// =============================================================================================
// backward, dK and dV of d = 128 in ONE pass (4 matrix products per query block instead of the split form's 5, one
// softmax / dropout recomputation instead of two).  256 keys per workgroup, 8 waves, two waves per SIMD = 256 registers:
//   * the wave's V rows stay in LDS (64 KB beside the 64 KB of Q / dO stages) and feed the dP MFMAs by ds_read_b128;
//     K fragments (32), dK (64) and dV (64) accumulators are the only long-lived registers;
//   * the tile loop is unrolled by two so the stage index is a compile-time constant: every LDS read is
//     (one of 8 row-fragment / 8 transposed-fragment address registers) + immediate, no address arithmetic in the loop;
//   * the dropout byte test is (word & bytemask) >= (thr << kbyte) on the quad-broadcast word (v_and_b32_dpp + v_cmp).
// Results are bit-identical to the split passes (same per-element arithmetic and accumulation order).
// =============================================================================================
template <bool DROP, bool RAGGED>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv128_kernel(const bf16_t* __restrict__ qkv,
                                                                 const bf16_t* __restrict__ dout,
                                                                 const float* __restrict__ lse,
                                                                 const float* __restrict__ delta,
                                                                 bf16_t* __restrict__ dqkv, int L, int H, float scale,
                                                                 unsigned thr, float dscale, uint64_t seed_arg, float opmul,
                                                                 float kgrad, int Lp, int ldq) {
  constexpr int D = 128, NW = 8;
  using C = Cfg<D>;
  constexpr int VOFF = 4 * C::TILE, SOFF = 8 * C::TILE;       // [2][Q|dO] | V rows of the workgroup | [2][lse2|delta|row hash]
  __shared__ __attribute__((aligned(16))) char smem[SOFF + 2 * 3 * 64 * 4];
  const uint64_t seed = seed_arg ^ o2_seed_salt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hq = lane >> 5;
  int tile_i, head, b;
  attn_tile_coords((L + NW * 32 - 1) / (NW * 32), H, tile_i, head, b);
  const int k0 = tile_i * (NW * 32) + wave * 32;
  const size_t tstride = (size_t)ldq;      // token-row pitch of qkv (and dqkv), >= 3 * H * D
  const size_t ostride = (size_t)H * D;
  const bf16_t* qbase = qkv + (size_t)b * L * tstride + (size_t)head * D;
  const bf16_t* kbase = qbase + (size_t)H * D;
  const bf16_t* vbase = qbase + (size_t)2 * H * D;
  const bf16_t* dobase = dout + (size_t)b * L * ostride + (size_t)head * D;
  const int krow_raw = k0 + (lane & 31);
  const bool k_ok = !RAGGED || krow_raw < L;
  const int krow = k_ok ? krow_raw : L - 1;
  const float sc_log2 = opmul;     // multiplier of the register-resident S operand: log2(e)/sqrt(d), or 1 when q is stored pre-scaled
  float* sstat = reinterpret_cast<float*>(smem + SOFF);

  bf16x8 kf[C::NDS];
#pragma unroll
  for (int ds = 0; ds < C::NDS; ++ds)
    kf[ds] = scale_frag(*reinterpret_cast<const bf16x8*>(kbase + (size_t)krow * tstride + ds * 16 + 8 * hq), sc_log2);
#pragma unroll
  for (int j = 0; j < NW / 2; ++j) {
    int start = tile_i * (NW * 32) + j * 64, nv = L - start;
    if (RAGGED && nv < 1) { start = L - 1; nv = 1; }      // a tile past the end reads the last row (its keys are discarded)
    stage64<D, RAGGED, NW>(vbase + (size_t)start * tstride, tstride, smem + VOFF + j * C::TILE, wave, lane, nv);
  }
  f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
  for (int i = 0; i < C::NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  // ---- LDS addresses: row fragments (k-step ds of row lane&31) and transposed fragments (head-dim block db)
  // (32-bit LDS byte addresses: one register each)
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  uint32_t rq[C::NDS];
  {
    const int r = lane & 31;
#pragma unroll
    for (int ds = 0; ds < C::NDS; ++ds) rq[ds] = lds0 + r * C::RB + (((ds * 2 + hq) ^ swz<D>(r)) << 4);
  }
  const int vwoff = VOFF + wave * 32 * C::RB;                 // this wave's 32 V rows
  uint32_t t0[C::NDB], t1[C::NDB];
  {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 4 * hq + q, row1 = row + 8;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) {
      const int c = db * 4 + 2 * (g & 1) + (p >> 1);
      t0[db] = lds0 + row * C::RB + ((c ^ swz<D>(row)) << 4) + 8 * (p & 1);
      t1[db] = lds0 + row1 * C::RB + ((c ^ swz<D>(row1)) << 4) + 8 * (p & 1);
    }
  }
  const uint32_t st4 = lds0 + SOFF + 16 * hq;                               // + ((buf*3 + which)*64 + qb*32 + 8*g4) * 4
  const uint32_t sh1 = lds0 + SOFF + (4 * hq + (lane & 3)) * 4;
  auto ld128 = [](uint32_t a_) { return *(const __attribute__((address_space(3))) bf16x8*)(a_); };
  auto ldf4 = [](uint32_t a_) { return *(const __attribute__((address_space(3))) f32x4*)(a_); };
  auto ldtr = [](uint32_t a0, uint32_t a1) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  const uint64_t bh = (uint64_t)(b * H + head);
  const uint32_t keyhash = DROP ? o2_attn_keyhash(seed, (uint32_t)(krow >> 2)) : 0u;   // this lane's key group
  const uint32_t kbyte = 8 * (krow & 3);
  const uint32_t bmask = 0xffu << kbyte, thrs = thr << kbyte;
  const int nt = (L + 63) / 64;
  const int nt2 = RAGGED ? ((nt + 1) & ~1) : nt;          // whole pairs of tiles (a tile past the end contributes nothing)
  // per-row statistics of a tile (-lse log2(e), -delta / dscale: the ready-made, padded tables of attn_delta_kernel) go
  // global -> LDS by two 256-byte LDS-DMA pieces, in the same vmcnt stream as the tile's own pieces: no register, no conversion
  // and -- unlike the load + convert + ds_write of rounds 1-2, whose first use of the loaded word sat at the top of the loop body
  // -- no wait for the load's latency and for every DMA piece issued before it, once per tile (round 3: -24 % on the d = 256
  // dK+dV kernel, profiles/r03_attn_dkv_stats_ab.txt).  The accumulators of S and dP start from these rows.
  const size_t sbase_p = ((size_t)(b * H + head)) * Lp;
  auto stage_stats = [&](int t, int buf) {
    if (wave < 2) {
      int ln = lane;
      asm volatile("" : "+v"(ln));        // the per-lane offset is rebuilt here, not kept in a register across the tile loop
      glds4_asm((wave ? delta : lse) + sbase_p + (size_t)t * 64, (uint32_t)ln * 4u,
                (uint32_t)(uintptr_t)LDS_PTR(float, sstat + (buf * 3 + wave) * 64));
    } else if (DROP && tid < 192) {   // dropout: hashes of the tile's 64 query rows
      int i = tid & 63;
      asm volatile("" : "+v"(i));         // (rebuilt here: hoisted out of the tile loop the slot address gets spilled, and its
                                          //  reload's vmcnt(0) drains the DMA pieces just issued)
      reinterpret_cast<uint32_t*>(sstat)[(buf * 3 + 2) * 64 + i] = o2_attn_rowhash(seed, bh * (uint64_t)L + (uint64_t)(t * 64 + i));
    }
  };
  auto stage_tile = [&](int t, int buf) {
    int start = t * 64, nv = L - start;
    if (RAGGED && nv < 1) { start = 0; nv = 64; }          // tile past the end: any valid rows (its statistics zero it)
    stage64<D, RAGGED, NW>(qbase + (size_t)start * tstride, tstride, smem + buf * 2 * C::TILE, wave, lane, nv);
    stage64<D, RAGGED, NW>(dobase + (size_t)start * ostride, ostride, smem + buf * 2 * C::TILE + C::TILE, wave, lane, nv);
    stage_stats(t, buf);
  };
  stage_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto body = [&](auto CURTAG, int t) {
    constexpr int CUR = decltype(CURTAG)::value;
    constexpr int QO = CUR * 2 * C::TILE, DOO = QO + C::TILE;
    if (t + 1 < nt2) stage_tile(t + 1, CUR ^ 1);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      // S[q x key] = Q . K^T, dP[q x key] = dO . V^T  (rows = queries in registers, column = key on the lane)
      // The accumulators START at -lse2[q] and -delta[q] (read from the tile's LDS table straight into them): with K
      // pre-multiplied by log2(e)/sqrt(d), S leaves the matrix pipe as the exp2 argument and dP as dP - delta.
      f32x16 s, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 a4 = ldf4(st4 + ((CUR * 3 + 0) * 64 + qb * 32 + 8 * g4) * 4);
        const f32x4 c4 = ldf4(st4 + ((CUR * 3 + 1) * 64 + qb * 32 + 8 * g4) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[4 * g4 + e] = a4[e]; dp[4 * g4 + e] = c4[e]; }
      }
      int vw = vwoff;
      asm volatile("" : "+s"(vw));          // keep the 8 V addresses out of registers: one v_add per read instead
      // one k-step of operands in flight (12 registers), pinned: hipcc otherwise prefetches all 24 fragments (96 registers)
      bf16x8 qf = ld128(rq[0] + QO + qb * 32 * C::RB);
      bf16x8 dof = ld128(rq[0] + DOO + qb * 32 * C::RB);
      bf16x8 vf = ld128(rq[0] + vw);
#pragma unroll
      for (int ds = 0; ds < C::NDS; ++ds) {
        bf16x8 qn = qf, don = dof, vn = vf;
        if (ds + 1 < C::NDS) {
          qn = ld128(rq[ds + 1] + QO + qb * 32 * C::RB);
          don = ld128(rq[ds + 1] + DOO + qb * 32 * C::RB);
          vn = ld128(rq[ds + 1] + vw);
        }
        s = MFMA32(qf, kf[ds], s);
        dp = MFMA32(dof, vf, dp);
        __builtin_amdgcn_sched_barrier(0);
        qf = qn; dof = don; vf = vn;
      }
      __builtin_amdgcn_sched_barrier(0);
      uint32_t hmine[4] = {0u, 0u, 0u, 0u};
      if (DROP) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) hmine[g4] = o2_attn_mix(*(const __attribute__((address_space(3))) uint32_t*)(sh1 + ((CUR * 3 + 2) * 64 + qb * 32 + 8 * g4) * 4), keyhash);
      }
      u32x4 pfw[2], dsw[2];     // P after dropout (for dV) and dS (for dK) as the operands of k-steps ss = 0, 1
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 ndl4 = {0.f, 0.f, 0.f, 0.f};
        if (DROP) ndl4 = ldf4(st4 + ((CUR * 3 + 1) * 64 + qb * 32 + 8 * g4) * 4);   // -delta
        float pv[4], dsv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g4 + e;
          const float p = __builtin_amdgcn_exp2f(s[r]);
          float dpr = dp[r];            // dP - delta
          float pdr = p;
          if (DROP) {
            const bool keep = (quad_bcast(hmine[g4], e) & bmask) >= thrs;
            dpr = keep ? dpr : ndl4[e];     // a dropped element: 0 - delta  (dscale is applied once, on the final dK / dV tiles)
            pdr = keep ? p : 0.f;
            asm volatile("" : "+v"(pdr));   // select in fp32, then convert pairs (else: single converts + v_perm merges)
          }
          pv[e] = pdr;
          dsv[e] = p * dpr;  // dS
        }
        pfw[g4 >> 1][2 * (g4 & 1)] = cvt_pk_bf2(pv[0], pv[1]);
        pfw[g4 >> 1][2 * (g4 & 1) + 1] = cvt_pk_bf2(pv[2], pv[3]);
        dsw[g4 >> 1][2 * (g4 & 1)] = cvt_pk_bf2(dsv[0], dsv[1]);
        dsw[g4 >> 1][2 * (g4 & 1) + 1] = cvt_pk_bf2(dsv[2], dsv[3]);
      }
      const bf16x8 pf[2] = {__builtin_bit_cast(bf16x8, pfw[0]), __builtin_bit_cast(bf16x8, pfw[1])};
      const bf16x8 dsf[2] = {__builtin_bit_cast(bf16x8, dsw[0]), __builtin_bit_cast(bf16x8, dsw[1])};
      __builtin_amdgcn_sched_barrier(0);
      {
        auto trd = [&](int j, int off) {     // j = ss*4 + db
          const int ro = (qb * 32 + (j >> 2) * 16) * C::RB + off;
          return ldtr(t0[j & 3] + ro, t1[j & 3] + ro);
        };
        bf16x8 fd = trd(0, DOO), fq = trd(0, QO);
#pragma unroll
        for (int j = 0; j < 2 * C::NDB; ++j) {
          bf16x8 fdn = fd, fqn = fq;
          if (j + 1 < 2 * C::NDB) { fdn = trd(j + 1, DOO); fqn = trd(j + 1, QO); }
          dv[j & 3] = MFMA32(fd, pf[j >> 2], dv[j & 3]);     // dV^T += dO^T . P
          dk[j & 3] = MFMA32(fq, dsf[j >> 2], dk[j & 3]);    // dK^T += Q^T . dS
          __builtin_amdgcn_sched_barrier(0);
          fd = fdn; fq = fqn;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int t = 0; t < nt2; t += 2) {
    body(IC<0>{}, t);
    body(IC<1>{}, t + 1);
  }
  if (!k_ok) return;
  bf16_t* dkrow = dqkv + ((size_t)b * L + krow) * tstride + (size_t)H * D + (size_t)head * D;
  bf16_t* dvrow = dkrow + (size_t)H * D;
  const float fk = DROP ? kgrad * dscale : kgrad, fv = DROP ? dscale : 1.0f;    // kgrad: 1/sqrt(d), or ln 2 (pre-scaled Q)
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dd = db * 32 + 8 * g4 + 4 * hq;
      u32x2 w;
      w[0] = pack_bf2(dk[db][4 * g4] * fk, dk[db][4 * g4 + 1] * fk);
      w[1] = pack_bf2(dk[db][4 * g4 + 2] * fk, dk[db][4 * g4 + 3] * fk);
      *reinterpret_cast<u32x2*>(dkrow + dd) = w;
      w[0] = pack_bf2(dv[db][4 * g4] * fv, dv[db][4 * g4 + 1] * fv);
      w[1] = pack_bf2(dv[db][4 * g4 + 2] * fv, dv[db][4 * g4 + 3] * fv);
      *reinterpret_cast<u32x2*>(dvrow + dd) = w;
    }
}

// =============================================================================================
// backward, dK and dV in one pass, d = 128, q stored pre-scaled, L % 256 == 0: the GENERATED kernel (tools/gen_attn_dkv.py)
// =============================================================================================
// One wave per SIMD; a wave owns 32 keys (dK^T, dV^T, its K and V fragments in the accumulator file), a workgroup 128 keys; the
// statement is the whole kernel (schedule: the generator's header; CPU execution of the same text: tests/test_attn_dkv_asm_emu_cpu.py).
// The compiler contributes the coordinates, the lane's key-group hash and the seed-dependent constant of the row hash
// (o2_hash64 with a zero high index word: requires B * H * L < 2^32, checked by the launcher).
template <bool DROP>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_w4_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                const float* __restrict__ nlse2, const float* __restrict__ ndelta,
                                                                bf16_t* __restrict__ dqkv, int L, int H, unsigned thr, float fk, float fv,
                                                                uint64_t seed_arg, int Lp, int ldq) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(1024))) char smem[O2_KV_LDS_BYTES];   // 4 x [Q 16 KiB] | 4 x [dO 16 KiB] | 4 x 1 KiB of row statistics
  const uint64_t seed = seed_arg ^ o2_seed_salt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_i, head, b;
  attn_tile_coords(L / 128, H, tile_i, head, b);
  const int k0 = tile_i * 128 + wave * 32;
  const size_t tstride = (size_t)ldq;      // token-row pitch of qkv (and dqkv), >= 3 * H * D
  const char* kptr = reinterpret_cast<const char*>(qkv + ((size_t)b * L + k0) * tstride + (size_t)H * D + (size_t)head * D);
  const char* vptr = kptr + (size_t)H * D * 2;
  const char* qptr = reinterpret_cast<const char*>(qkv + (size_t)b * L * tstride + (size_t)head * D);
  const char* doptr = reinterpret_cast<const char*>(dout + ((size_t)b * L * H + head) * D);
  const char* lseptr = reinterpret_cast<const char*>(nlse2 + ((size_t)(b * H + head)) * Lp);
  const char* dltptr = reinterpret_cast<const char*>(ndelta + ((size_t)(b * H + head)) * Lp);
  char* okptr = reinterpret_cast<char*>(dqkv + ((size_t)b * L + k0) * tstride + (size_t)H * D + (size_t)head * D);
  char* ovptr = okptr + (size_t)H * D * 2;
  const uint32_t keyh = DROP ? o2_attn_keyhash(seed, (uint32_t)((k0 + (lane & 31)) >> 2)) : 0u;
  const uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
  const uint32_t hseed = s_lo ^ ((s_hi << 16) | (s_hi >> 16)) ^ (s_hi + (s_hi << 3));     // o2_hash64(seed, idx) = mix(idx ^ hseed) for idx < 2^32
  const uint32_t rowbase = (uint32_t)((uint64_t)(b * H + head) * (uint64_t)L);
  const uint32_t ldsb = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const int nt = L / 64, strideb = (int)(tstride * 2), dorowb = H * D * 2;
  const uint32_t fkb = __float_as_uint(fk), fvb = __float_as_uint(fv);
#define O2_KV_OPERANDS                                                                                                              \
  [kptr] "s"(kptr), [vptr] "s"(vptr), [qptr] "s"(qptr), [doptr] "s"(doptr), [lseptr] "s"(lseptr), [dltptr] "s"(dltptr),            \
      [okptr] "s"(okptr), [ovptr] "s"(ovptr), [nt] "s"(nt), [strideb] "s"(strideb), [dorowb] "s"(dorowb), [ldsb] "s"(ldsb),        \
      [wave] "s"(wave), [thr] "s"(thr), [fk] "s"(fkb), [fv] "s"(fvb), [rowbase] "s"(rowbase), [hseed] "s"(hseed), [keyh] "v"(keyh)
  if constexpr (DROP) {
    asm volatile(O2_KV_ASM_DROP : : O2_KV_OPERANDS : O2_KV_CLOBBERS);
  } else {
    asm volatile(O2_KV_ASM_NODROP : : O2_KV_OPERANDS : O2_KV_CLOBBERS);
  }
#undef O2_KV_OPERANDS
}

// =============================================================================================
// backward, dK and dV of d = 256 (interm_10b) in ONE pass: the construction of attn_bwd_dkv128_kernel at one wave per SIMD.
// 128 keys per workgroup (4 waves x 32), 512 registers per wave: dK (128) + dV (128) accumulators, K fragments (64), the
// wave's V rows in LDS (64 KB for the workgroup) feeding the dP MFMAs by ds_read_b128.  The LDS holds that plus two stages of
// 32-ROW Q / dO tiles (2 x 2 x 16 KB): a tile = one query block = 64 MFMAs per wave (S 16, dP 16, dV 16, dK 16) against the
// two-pass form's 48 + 32 + a second softmax / dropout recomputation.  Same per-element arithmetic and accumulation order as the
// split passes: bit-identical dK / dV (tests/test_hip_ops.py::test_attention_dkv_one_pass_equals_two_passes).
//   * row fragments of a 512-byte row: chunk c = 2 ds + h, the swizzle touches its low 4 bits only, so k-steps ds and ds + 8 are
//     one address register + 256 (8 registers serve 16 k-steps); transposed blocks db and db + 4 likewise (8 registers serve 8);
//   * the tile loop is unrolled by two (compile-time stage index: every LDS address is a register + immediate);
//   * one k-step of operands in flight, pinned by sched_barrier (as in the d = 128 kernel).
// =============================================================================================
template <bool DROP, bool RAGGED>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv256_kernel(const bf16_t* __restrict__ qkv,
                                                                 const bf16_t* __restrict__ dout,
                                                                 const float* __restrict__ lse,
                                                                 const float* __restrict__ delta,
                                                                 bf16_t* __restrict__ dqkv, int L, int H, float scale,
                                                                 unsigned thr, float dscale, uint64_t seed_arg, float opmul,
                                                                 float kgrad, int Lp, int ldq) {
  constexpr int D = 256, NW = 4, TR = 32;                    // TR: query rows per staged tile
  using C = Cfg<D>;
  constexpr int TILE = TR * C::RB;                           // 16 KB
  constexpr int VOFF = 4 * TILE, SOFF = VOFF + NW * 32 * C::RB;      // [2][Q|dO] | V rows of the workgroup | [2][lse2|delta|row hash]
  __shared__ __attribute__((aligned(16))) char smem[SOFF + 2 * 3 * TR * 4];
  const uint64_t seed = seed_arg ^ o2_seed_salt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hq = lane >> 5;
  int tile_i, head, b;
  attn_tile_coords((L + NW * 32 - 1) / (NW * 32), H, tile_i, head, b);
  const int k0 = tile_i * (NW * 32) + wave * 32;
  const size_t tstride = (size_t)ldq;      // token-row pitch of qkv (and dqkv), >= 3 * H * D
  const size_t ostride = (size_t)H * D;
  const bf16_t* qbase = qkv + (size_t)b * L * tstride + (size_t)head * D;
  const bf16_t* kbase = qbase + (size_t)H * D;
  const bf16_t* vbase = qbase + (size_t)2 * H * D;
  const bf16_t* dobase = dout + (size_t)b * L * ostride + (size_t)head * D;
  const int krow_raw = k0 + (lane & 31);
  const bool k_ok = !RAGGED || krow_raw < L;
  const int krow = k_ok ? krow_raw : L - 1;
  float* sstat = reinterpret_cast<float*>(smem + SOFF);

  // stage ROWS rows (a multiple of 2 * NW) of a [rows][256] bf16 matrix: 1-KiB pieces of two rows each, swizzled source
  // (inline-asm LDS-DMA, common.h: the compiler drains the builtin form in the middle of the tile)
  auto stage_rows = [&](const bf16_t* base, size_t stride, char* tile, int rows, int nvalid) {
    const int pieces = rows / 2 / NW;                        // per wave
    const uint32_t dst0 = (uint32_t)(uintptr_t)LDS_PTR(char, tile);
    for (int t = 0; t < pieces; ++t) {
      const int i = wave * pieces + t;
      const int row = i * 2 + lane / 32;
      const int c = (lane % 32) ^ swz<D>(row);
      const int rsrc = (!RAGGED || row < nvalid) ? row : nvalid - 1;
      const uint32_t off = ((uint32_t)rsrc * (uint32_t)stride + (uint32_t)(c * 8)) * 2u;
#ifdef O2_DKV256_BUILTIN_DMA
      glds16(reinterpret_cast<const char*>(base) + off, tile + i * 1024);
#else
      glds16_asm(base, off, dst0 + i * 1024);
#endif
    }
  };

  bf16x8 kf[C::NDS];
#pragma unroll
  for (int ds = 0; ds < C::NDS; ++ds)
    kf[ds] = scale_frag(*reinterpret_cast<const bf16x8*>(kbase + (size_t)krow * tstride + ds * 16 + 8 * hq), opmul);
  {
    int start = tile_i * (NW * 32), nv = L - start;
    if (RAGGED && nv < 1) { start = L - 1; nv = 1; }
    stage_rows(vbase + (size_t)start * tstride, tstride, smem + VOFF, NW * 32, nv);
  }
  f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
  for (int i = 0; i < C::NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  // ---- LDS addresses (32-bit): row fragments of row lane&31 for k-steps 0..7 (+256: k-steps 8..15), transposed blocks 0..3 (+256: 4..7)
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  uint32_t rq[8];
  {
    const int r = lane & 31;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) rq[ds] = lds0 + r * C::RB + (((ds * 2 + hq) ^ swz<D>(r)) << 4);
  }
  const uint32_t vwoff = VOFF + wave * 32 * C::RB;           // this wave's 32 V rows
  uint32_t t0[4], t1[4];
  {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 4 * hq + q, row1 = row + 8;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int c = db * 4 + 2 * (g & 1) + (p >> 1);
      t0[db] = lds0 + row * C::RB + ((c ^ swz<D>(row)) << 4) + 8 * (p & 1);
      t1[db] = lds0 + row1 * C::RB + ((c ^ swz<D>(row1)) << 4) + 8 * (p & 1);
    }
  }
  auto ld128 = [](uint32_t a) { return *(const __attribute__((address_space(3))) bf16x8*)(a); };
  auto ldf4 = [](uint32_t a) { return *(const __attribute__((address_space(3))) f32x4*)(a); };
  auto ldtr = [](uint32_t a0, uint32_t a1) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  const uint32_t st4 = lds0 + SOFF + 16 * hq;                               // + ((buf*3 + which)*TR + 8*g4) * 4
  const uint32_t sh1 = lds0 + SOFF + (4 * hq + (lane & 3)) * 4;

  const uint64_t bh = (uint64_t)(b * H + head);
  const uint32_t keyhash = DROP ? o2_attn_keyhash(seed, (uint32_t)(krow >> 2)) : 0u;   // this lane's key group
  const uint32_t kbyte = 8 * (krow & 3);
  const uint32_t bmask = 0xffu << kbyte, thrs = thr << kbyte;
  const int nt = (L + TR - 1) / TR;
  const int nt2 = (nt + 1) & ~1;                            // whole pairs of tiles (a tile past the end contributes nothing)
  // per-row statistics (see attn_bwd_dkv128_kernel): two LDS-DMA pieces of TR dwords from the ready-made, padded tables
  const size_t sbase_p = ((size_t)(b * H + head)) * Lp;
  auto stage_stats = [&](int t, int buf) {
    if (wave < 2) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      if (ln < TR)
        glds4_asm((wave ? delta : lse) + sbase_p + (size_t)t * TR, (uint32_t)ln * 4u,
                  (uint32_t)(uintptr_t)LDS_PTR(float, sstat + (buf * 3 + wave) * TR));
    } else if (DROP && tid < 128 + TR) {   // dropout: hashes of the tile's query rows
      int i = tid - 128;
      asm volatile("" : "+v"(i));
      reinterpret_cast<uint32_t*>(sstat)[(buf * 3 + 2) * TR + i] = o2_attn_rowhash(seed, bh * (uint64_t)L + (uint64_t)(t * TR + i));
    }
  };
  auto stage_tile = [&](int t, int buf) {
    int start = t * TR, nv = L - start;
    if (nv < 1) { start = 0; nv = L < TR ? L : TR; }       // tile past the end: any valid rows (its statistics zero it)
    stage_rows(qbase + (size_t)start * tstride, tstride, smem + buf * 2 * TILE, TR, nv);
    stage_rows(dobase + (size_t)start * ostride, ostride, smem + buf * 2 * TILE + TILE, TR, nv);
    stage_stats(t, buf);
  };
  stage_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto body = [&](auto CURTAG, int t) {
    constexpr int CUR = decltype(CURTAG)::value;
    constexpr int QO = CUR * 2 * TILE, DOO = QO + TILE;
    if (t + 1 < nt2) stage_tile(t + 1, CUR ^ 1);
    // S[q x key] = Q . K^T - lse2, dP[q x key] = dO . V^T - delta  (rows = queries in registers, column = key on the lane)
    f32x16 s, dp;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 a4 = ldf4(st4 + ((CUR * 3 + 0) * TR + 8 * g4) * 4);
      const f32x4 c4 = ldf4(st4 + ((CUR * 3 + 1) * TR + 8 * g4) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[4 * g4 + e] = a4[e]; dp[4 * g4 + e] = c4[e]; }
    }
    uint32_t vw = vwoff;
    asm volatile("" : "+s"(vw));          // keep the V addresses out of registers: one v_add per read instead
    auto rowaddr = [&](int ds) { return rq[ds & 7] + (ds >> 3) * 256; };
    // O2_LA k-steps of operands in flight (one wave per SIMD: nobody else covers an LDS read's latency), pinned
    constexpr int LA = O2_DKV256_LA;
    bf16x8 qr[LA + 1], dr[LA + 1], vr[LA + 1];
#pragma unroll
    for (int i = 0; i < LA; ++i) { qr[i] = ld128(rowaddr(i) + QO); dr[i] = ld128(rowaddr(i) + DOO); vr[i] = ld128(rowaddr(i) + vw); }
#pragma unroll
    for (int ds = 0; ds < C::NDS; ++ds) {
      if (ds + LA < C::NDS) {
        constexpr int dummy = 0; (void)dummy;
        qr[(ds + LA) % (LA + 1)] = ld128(rowaddr(ds + LA) + QO);
        dr[(ds + LA) % (LA + 1)] = ld128(rowaddr(ds + LA) + DOO);
        vr[(ds + LA) % (LA + 1)] = ld128(rowaddr(ds + LA) + vw);
      }
      s = MFMA32(qr[ds % (LA + 1)], kf[ds], s);
      dp = MFMA32(dr[ds % (LA + 1)], vr[ds % (LA + 1)], dp);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    uint32_t hmine[4] = {0u, 0u, 0u, 0u};
    if (DROP) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        hmine[g4] = o2_attn_mix(*(const __attribute__((address_space(3))) uint32_t*)(sh1 + ((CUR * 3 + 2) * TR + 8 * g4) * 4), keyhash);
    }
    u32x4 pfw[2], dsw[2];     // P after dropout (for dV) and dS (for dK) as the operands of k-steps ss = 0, 1
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 ndl4 = {0.f, 0.f, 0.f, 0.f};
      if (DROP) ndl4 = ldf4(st4 + ((CUR * 3 + 1) * TR + 8 * g4) * 4);   // -delta
      float pv[4], dsv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        const float p = __builtin_amdgcn_exp2f(s[r]);
        float dpr = dp[r];            // dP - delta
        float pdr = p;
        if (DROP) {
          const bool keep = (quad_bcast(hmine[g4], e) & bmask) >= thrs;
          dpr = keep ? dpr : ndl4[e];     // a dropped element: 0 - delta  (dscale is applied once, on the final dK / dV tiles)
          pdr = keep ? p : 0.f;
          asm volatile("" : "+v"(pdr));   // select in fp32, then convert pairs
        }
        pv[e] = pdr;
        dsv[e] = p * dpr;  // dS
      }
      pfw[g4 >> 1][2 * (g4 & 1)] = cvt_pk_bf2(pv[0], pv[1]);
      pfw[g4 >> 1][2 * (g4 & 1) + 1] = cvt_pk_bf2(pv[2], pv[3]);
      dsw[g4 >> 1][2 * (g4 & 1)] = cvt_pk_bf2(dsv[0], dsv[1]);
      dsw[g4 >> 1][2 * (g4 & 1) + 1] = cvt_pk_bf2(dsv[2], dsv[3]);
    }
    const bf16x8 pf[2] = {__builtin_bit_cast(bf16x8, pfw[0]), __builtin_bit_cast(bf16x8, pfw[1])};
    const bf16x8 dsf[2] = {__builtin_bit_cast(bf16x8, dsw[0]), __builtin_bit_cast(bf16x8, dsw[1])};
    __builtin_amdgcn_sched_barrier(0);
    {
      auto trd = [&](int j, int off) {     // j = ss * NDB + db
        const int db = j % C::NDB, ss = j / C::NDB;
        const uint32_t ro = (ss * 16) * C::RB + off + (db >> 2) * 256;
        return ldtr(t0[db & 3] + ro, t1[db & 3] + ro);
      };
      constexpr int LB = O2_DKV256_LA;
      bf16x8 fdr[LB + 1], fqr[LB + 1];
#pragma unroll
      for (int i = 0; i < LB; ++i) { fdr[i] = trd(i, DOO); fqr[i] = trd(i, QO); }
#pragma unroll
      for (int j = 0; j < 2 * C::NDB; ++j) {
        if (j + LB < 2 * C::NDB) { fdr[(j + LB) % (LB + 1)] = trd(j + LB, DOO); fqr[(j + LB) % (LB + 1)] = trd(j + LB, QO); }
        dv[j % C::NDB] = MFMA32(fdr[j % (LB + 1)], pf[j / C::NDB], dv[j % C::NDB]);     // dV^T += dO^T . P
        dk[j % C::NDB] = MFMA32(fqr[j % (LB + 1)], dsf[j / C::NDB], dk[j % C::NDB]);    // dK^T += Q^T . dS
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int t = 0; t < nt2; t += 2) {
    body(IC<0>{}, t);
    body(IC<1>{}, t + 1);
  }
  if (!k_ok) return;
  bf16_t* dkrow = dqkv + ((size_t)b * L + krow) * tstride + (size_t)H * D + (size_t)head * D;
  bf16_t* dvrow = dkrow + (size_t)H * D;
  const float fk = DROP ? kgrad * dscale : kgrad, fv = DROP ? dscale : 1.0f;
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dd = db * 32 + 8 * g4 + 4 * hq;
      u32x2 w;
      w[0] = pack_bf2(dk[db][4 * g4] * fk, dk[db][4 * g4 + 1] * fk);
      w[1] = pack_bf2(dk[db][4 * g4 + 2] * fk, dk[db][4 * g4 + 3] * fk);
      *reinterpret_cast<u32x2*>(dkrow + dd) = w;
      w[0] = pack_bf2(dv[db][4 * g4] * fv, dv[db][4 * g4 + 1] * fv);
      w[1] = pack_bf2(dv[db][4 * g4 + 2] * fv, dv[db][4 * g4 + 3] * fv);
      *reinterpret_cast<u32x2*>(dvrow + dd) = w;
    }
}

}  // namespace

O2_DEFINE_SALT_OP(attn)

#ifdef O2_STAMP
extern "C" int orbit2_debug_read_attn(unsigned int* host_dst, int n) {   // diagnostic build only
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(o2_dbg_attn), sizeof(unsigned) * (size_t)n);
}
#endif

static int attn_check(const void* a, const void* b, int B, int L, int H, int d, float p) {
  if (!a || !b || B <= 0 || L <= 0 || H <= 0) return O2_ERR_ARG;
  if (d != 64 && d != 128 && d != 256) return O2_ERR_UNSUPPORTED;
  if (p < 0.f || p >= 1.f) return O2_ERR_ARG;
  return O2_OK;
}

// waves per workgroup: 8 (256-row tiles, K/V or Q/dO tiles shared by twice the waves) whenever the sequence has at least
// one such tile and the kernel fits two waves per SIMD (d = 64, 128); ORBIT2_ATTN_4WAVES keeps the round-1 geometry (A/B).
// The variant is an ARGUMENT of the *_ex entry points (no process-global switch: a forward / backward pair cannot disagree
// behind the caller's back, nothing is read from the environment on the launch path).
static int attn_waves(int L, int d, int flags) {
  if (d == 256 || L < 256) return 4;
  return (flags & ORBIT2_ATTN_4WAVES) ? 4 : 8;
}
// the FORWARD at d = 64 is faster with 4-wave workgroups, two per CU (round 3, same-box: -5 % without, -12 % with dropout at
// L = 4096; the backward is neutral, d = 128 gains 2.5-6 % / 21-23 % from 8 waves): forward and backward pick independently
static int attn_waves_fwd(int L, int d, int flags) {
  return d == 64 ? 4 : attn_waves(L, d, flags);
}

template <int DV, bool DR, bool RG, int NW>
static void launch_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, float sc_log2, unsigned thr, float dscale,
                       uint64_t seed, hipStream_t s, int ldo, int ldq) {
  dim3 grid(((L + NW * 32 - 1) / (NW * 32)) * H * B), block(NW * 64);
  if constexpr (!DR) {
    hipLaunchKernelGGL((attn_fwd_lazy_kernel<DV, RG, NW>), grid, block, 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, sc_log2,
                       thr, dscale, seed, ldo, ldq);
    return;
  }
  hipLaunchKernelGGL((attn_fwd_kernel<DV, DR, RG, NW>), grid, block, 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, sc_log2,
                     thr, dscale, seed, ldo, ldq);
}
template <int DV, bool DR, int NW>
static void launch_fwd_r(bool ragged, const void* qkv, void* out, float* lse, int B, int L, int H, float sc_log2, unsigned thr,
                         float dscale, uint64_t seed, hipStream_t s, int ldo, int ldq) {
  if (ragged) launch_fwd<DV, DR, true, NW>(qkv, out, lse, B, L, H, sc_log2, thr, dscale, seed, s, ldo, ldq);
  else launch_fwd<DV, DR, false, NW>(qkv, out, lse, B, L, H, sc_log2, thr, dscale, seed, s, ldo, ldq);
}

extern "C" int orbit2_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int d, float drop_p,
                               uint64_t seed, void* stream) {
  return orbit2_attn_fwd_ex(qkv, out, lse, B, L, H, d, drop_p, seed, 0, stream);
}

extern "C" int orbit2_attn_fwd_ex(const void* qkv, void* out, float* lse, int B, int L, int H, int d, float drop_p,
                                  uint64_t seed, int flags, void* stream) {
  return orbit2_attn_fwd_ld(qkv, out, lse, B, L, H, d, drop_p, seed, flags, 3 * H * d, H * d, stream);
}

extern "C" int orbit2_attn_fwd_ld(const void* qkv, void* out, float* lse, int B, int L, int H, int d, float drop_p,
                                  uint64_t seed, int flags, int ldq, int ldo, void* stream) {
  int rc = attn_check(qkv, out, B, L, H, d, drop_p);
  if (rc) return rc;
  if (!lse || ldo < H * d || (ldo & 7) || ldq < 3 * H * d || (ldq & 7)) return O2_ERR_ARG;
  // (q stored pre-scaled: the kernels' multiplier is 1)
  const float sc_log2 = (flags & ORBIT2_ATTN_Q_PRESCALED) ? 1.0f : (1.0f / sqrtf((float)d)) * 1.4426950408889634f;
  const unsigned thr = (unsigned)(drop_p * 256.0f + 0.5f);
  const float dscale = 256.0f / (256.0f - (float)thr);
  hipStream_t s = (hipStream_t)stream;
  if (d == 128 && (flags & ORBIT2_ATTN_Q_PRESCALED) && !(flags & ORBIT2_ATTN_NO_W4) && L % 256 == 0 && L <= O2_AF_MAX_L &&
      attn_w4_range_ok(L, ldq, ldo)) {
    dim3 grid((unsigned)((L / 256) * H * B)), block(256);
    if (thr) hipLaunchKernelGGL((attn_fwd_w4_kernel<true>), grid, block, 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, thr, dscale, seed, ldo, ldq);
    else hipLaunchKernelGGL((attn_fwd_w4_kernel<false>), grid, block, 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, thr, dscale, seed, ldo, ldq);
    O2_CHECK_LAUNCH();
    return O2_OK;
  }
  const int nw = attn_waves_fwd(L, d, flags);
  const bool ragged = (L % (nw * 32)) != 0;
#define O2_FWD(DV, NWV)                                                                              \
  do {                                                                                               \
    if (thr) launch_fwd_r<DV, true, NWV>(ragged, qkv, out, lse, B, L, H, sc_log2, thr, dscale, seed, s, ldo, ldq);  \
    else launch_fwd_r<DV, false, NWV>(ragged, qkv, out, lse, B, L, H, sc_log2, thr, dscale, seed, s, ldo, ldq);     \
  } while (0)
  if (d == 256) O2_FWD(256, 4);
  else if (d == 128) { if (nw == 8) O2_FWD(128, 8); else O2_FWD(128, 4); }
  else { if (nw == 8) O2_FWD(64, 8); else O2_FWD(64, 4); }
#undef O2_FWD
  O2_CHECK_LAUNCH();
  return O2_OK;
}

template <int DV, bool DR, bool RG, int NW>
static void launch_bwd(const bf16_t* q_, const bf16_t* do_, const float* lse, const float* delta, bf16_t* dq_, int B, int L,
                       int H, float scale, unsigned thr, float dscale, uint64_t seed, hipStream_t s, int flags, int Lp, int ldq) {
  dim3 grid(((L + NW * 32 - 1) / (NW * 32)) * H * B), block(NW * 64);
  const bool pre = (flags & ORBIT2_ATTN_Q_PRESCALED) != 0;
  const float opmul = pre ? 1.0f : scale * 1.4426950408889634f, kgrad = pre ? 0.6931471805599453f : scale;
  const bool w4_range = attn_w4_range_ok(L, ldq, ldq);     // (dO / dqkv rows are no wider than the qkv pitch)
  if (DV == 128 && pre && !(flags & ORBIT2_ATTN_NO_W4) && L % 256 == 0 && L <= O2_AF_MAX_L && w4_range) {
    // the generated one-wave-per-SIMD dQ kernel (256-row workgroups whatever NW is)
    hipLaunchKernelGGL((attn_bwd_dq_w4_kernel<DR>), dim3((unsigned)((L / 256) * H * B)), dim3(256), 0, s, q_, do_, lse, delta, dq_,
                       L, H, thr, DR ? scale * dscale : scale, seed, Lp, ldq);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DV, DR, RG, NW>), grid, block, 0, s, q_, do_, lse, delta, dq_, L, H, scale, thr, dscale,
                       seed, opmul, Lp, ldq);
  }
  if (DV == 128 && pre && !(flags & (ORBIT2_ATTN_NO_W4 | ORBIT2_ATTN_SPLIT_DKV)) && L % 256 == 0 && L <= O2_AF_MAX_L &&
      (uint64_t)B * (uint64_t)H * (uint64_t)L < (1ull << 32) && w4_range) {
    // the generated one-wave-per-SIMD dK + dV kernel: 128 keys per workgroup
    hipLaunchKernelGGL((attn_bwd_dkv_w4_kernel<DR>), dim3((unsigned)((L / 128) * H * B)), dim3(256), 0, s, q_, do_, lse, delta, dq_,
                       L, H, thr, DR ? kgrad * dscale : kgrad, DR ? dscale : 1.0f, seed, Lp, ldq);
    return;
  }
  if constexpr (DV == 64) {       // dK and dV in one pass (fits two waves per SIMD)
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DV, DR, 0, RG, NW>), grid, block, 0, s, q_, do_, lse, delta, dq_, L, H, scale, thr,
                       dscale, seed, opmul, kgrad, Lp, ldq);
  } else if (DV == 256 && !(flags & ORBIT2_ATTN_SPLIT_DKV)) {             // one pass at one wave per SIMD (interm_10b)
    hipLaunchKernelGGL((attn_bwd_dkv256_kernel<DR, RG>), grid, block, 0, s, q_, do_, lse, delta, dq_, L, H, scale, thr, dscale,
                       seed, opmul, kgrad, Lp, ldq);
  } else if (DV == 128 && NW == 8 && !(flags & ORBIT2_ATTN_SPLIT_DKV)) {   // one pass, V rows in LDS
    hipLaunchKernelGGL((attn_bwd_dkv128_kernel<DR, RG>), grid, block, 0, s, q_, do_, lse, delta, dq_, L, H, scale, thr, dscale,
                       seed, opmul, kgrad, Lp, ldq);
  } else {
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DV, DR, 1, RG, NW>), grid, block, 0, s, q_, do_, lse, delta, dq_, L, H, scale, thr,
                       dscale, seed, opmul, kgrad, Lp, ldq);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DV, DR, 2, RG, NW>), grid, block, 0, s, q_, do_, lse, delta, dq_, L, H, scale, thr,
                       dscale, seed, opmul, kgrad, Lp, ldq);
  }
}
template <int DV, int NW>
static void launch_bwd_r(bool drop, bool ragged, const bf16_t* q_, const bf16_t* do_, const float* lse, const float* delta,
                         bf16_t* dq_, int B, int L, int H, float scale, unsigned thr, float dscale, uint64_t seed, hipStream_t s,
                         int flags, int Lp, int ldq) {
  if (drop) {
    if (ragged) launch_bwd<DV, true, true, NW>(q_, do_, lse, delta, dq_, B, L, H, scale, thr, dscale, seed, s, flags, Lp, ldq);
    else launch_bwd<DV, true, false, NW>(q_, do_, lse, delta, dq_, B, L, H, scale, thr, dscale, seed, s, flags, Lp, ldq);
  } else {
    if (ragged) launch_bwd<DV, false, true, NW>(q_, do_, lse, delta, dq_, B, L, H, scale, thr, dscale, seed, s, flags, Lp, ldq);
    else launch_bwd<DV, false, false, NW>(q_, do_, lse, delta, dq_, B, L, H, scale, thr, dscale, seed, s, flags, Lp, ldq);
  }
}

static int attn_lpad(int L) { return ((L + 63) / 64) * 64 + 64; }     // padded row stride of the statistics tables

extern "C" int64_t orbit2_attn_bwd_ws_floats(int B, int L, int H) {
  if (B <= 0 || L <= 0 || H <= 0) return 0;
  return (int64_t)2 * B * H * attn_lpad(L);
}

extern "C" int orbit2_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                               void* dqkv, int B, int L, int H, int d, float drop_p, uint64_t seed, void* stream) {
  return orbit2_attn_bwd_ex(qkv, out, dout, lse, delta, dqkv, B, L, H, d, drop_p, seed, 0, stream);
}

extern "C" int orbit2_attn_bwd_ex(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                                  void* dqkv, int B, int L, int H, int d, float drop_p, uint64_t seed, int flags,
                                  void* stream) {
  return orbit2_attn_bwd_ld(qkv, out, dout, lse, delta, dqkv, B, L, H, d, drop_p, seed, flags, 3 * H * d, H * d, stream);
}

extern "C" int orbit2_attn_bwd_ld(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                                  void* dqkv, int B, int L, int H, int d, float drop_p, uint64_t seed, int flags, int ldq,
                                  int ldo, void* stream) {
  int rc = attn_check(qkv, out, B, L, H, d, drop_p);
  if (rc) return rc;
  if (!dout || !lse || !delta || !dqkv || ldo < H * d || (ldo & 7) || ldq < 3 * H * d || (ldq & 7)) return O2_ERR_ARG;
  const float scale = 1.0f / sqrtf((float)d);
  const unsigned thr = (unsigned)(drop_p * 256.0f + 0.5f);
  const float dscale = 256.0f / (256.0f - (float)thr);
  hipStream_t s = (hipStream_t)stream;
  // per-row statistics tables (attn_delta_kernel): delta = workspace of orbit2_attn_bwd_ws_floats(B, L, H) floats
  const int Lp = attn_lpad(L);
  float* ws0 = delta;
  float* ws1 = delta + (size_t)B * H * Lp;
  const int64_t nrows = (int64_t)B * Lp * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nrows * 16 + 255) / 256)), dim3(256), 0, s,
                     (const bf16_t*)out, (const bf16_t*)dout, lse, delta, B, L, H, d, Lp, 1.0f / dscale, ldo);
  O2_CHECK_LAUNCH();
  const bf16_t* q_ = (const bf16_t*)qkv;
  const bf16_t* do_ = (const bf16_t*)dout;
  bf16_t* dq_ = (bf16_t*)dqkv;
  const int nw = attn_waves(L, d, flags);
  const bool ragged = (L % (nw * 32)) != 0;
  if (d == 256) launch_bwd_r<256, 4>(thr != 0, ragged, q_, do_, ws0, ws1, dq_, B, L, H, scale, thr, dscale, seed, s, flags, Lp, ldq);
  else if (d == 128) {
    if (nw == 8) launch_bwd_r<128, 8>(thr != 0, ragged, q_, do_, ws0, ws1, dq_, B, L, H, scale, thr, dscale, seed, s, flags, Lp, ldq);
    else launch_bwd_r<128, 4>(thr != 0, ragged, q_, do_, ws0, ws1, dq_, B, L, H, scale, thr, dscale, seed, s, flags, Lp, ldq);
  } else {
    if (nw == 8) launch_bwd_r<64, 8>(thr != 0, ragged, q_, do_, ws0, ws1, dq_, B, L, H, scale, thr, dscale, seed, s, flags, Lp, ldq);
    else launch_bwd_r<64, 4>(thr != 0, ragged, q_, do_, ws0, ws1, dq_, B, L, H, scale, thr, dscale, seed, s, flags, Lp, ldq);
  }
  O2_CHECK_LAUNCH();
  return O2_OK;
}
