// bf16 MFMA GEMM for gfx950 with a fused epilogue.  See include/orbit2_hip.h (orbit2_gemm_bf16).
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 fragments of
// v_mfma_f32_16x16x32_bf16.  Operands are staged global->LDS by LDS-DMA (global_load_lds_dwordx4,
// 1 KiB per wave-instruction, lane-linear destination); the bank-conflict swizzle is applied on the
// per-lane SOURCE address and undone on the fragment read:
//   K-contiguous operand  : LDS image [128 rows][64 k]   (128-B rows), 16-B chunk ^= row&7,
//                           fragments by ds_read_b128.
//   K-strided operand     : LDS image [64 k][128 cols]   (256-B rows), 32-B granule ^= (k&3)|((k>>3)&1)<<2,
//                           fragments by 2x ds_read_b64_tr_b16 (hardware transpose).
// The MFMA is issued with the N-side fragment as the A operand and the M-side fragment as B, so a lane
// ends up holding 4 consecutive n of one output row m -> 8-byte bf16 (16-byte fp32) row-major stores.
#include "common.h"
#include "../../include/orbit2_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand per stage

// 16 zero bytes every lane may fetch: the source of LDS-DMA pieces that lie past the end of the contraction (K tail)
__device__ __attribute__((aligned(16))) const unsigned int o2_zero16[4] = {0u, 0u, 0u, 0u};

template <bool KC, int NP = 4>   // NP pieces per wave: 4 = a 128-row tile; 2 = the 64-row tile of a K-contiguous operand
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int ld, int r0, int rmax, int k0, int kmax,
                                           char* tile, int wave, int lane) {
  static_assert(NP == 4 || KC, "the 64-row tile exists for K-contiguous operands only");
  const bool tail = k0 + BK > kmax;   // wave-uniform: only the last k-step of a ragged K pays the per-lane redirect
#pragma unroll
  for (int t = 0; t < NP; ++t) {
    const int i = wave * NP + t;  // LDS-DMA instruction id, 0..15 (1 KiB each)
    const bf16_t* src;
    if (KC) {
      const int row = i * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (row & 7);
      int gr = r0 + row;
      gr = gr < rmax ? gr : rmax - 1;
      src = G + (size_t)gr * ld + k0 + chunk * 8;
      if (tail && k0 + chunk * 8 >= kmax) src = reinterpret_cast<const bf16_t*>(o2_zero16);
    } else {
      const int krow = i * 4 + (lane >> 4);
      const int cp = lane & 15;
      const int sw = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int chunk = ((((cp >> 1) ^ sw)) << 1) | (cp & 1);
      int col = r0 + chunk * 8;
      col = col <= rmax - 8 ? col : rmax - 8;
      src = G + (size_t)(k0 + krow) * ld + col;
      if (tail && k0 + krow >= kmax) src = reinterpret_cast<const bf16_t*>(o2_zero16);
    }
    glds16(src, tile + i * 1024);
  }
}

template <bool KC>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int r0, int kk, int lane) {
  if (KC) {
    const int row = r0 + (lane & 15);
    const int chunk = kk * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int krow = kk * 32 + 8 * g + q;
    const int sw = q | ((g & 1) << 2);
    const char* a = tile + krow * 256 + ((((r0 >> 4) ^ sw)) << 5) + 8 * pp;
    const bf16x4 lo = lds_tr4(a);
    const bf16x4 hi = lds_tr4(a + 4 * 256);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

struct Epi {
  const bf16_t* bias;
  bf16_t* save_pre;
  const bf16_t* dgelu_pre;
  const float* rowscale;
  const bf16_t* residual;
  void* C;
  uint64_t seed;
  int M, N, ldc, ldr, res_mod, res_first, rows_per_scale, act, out_fp32;
  unsigned thr;
  float dscale, beta;
  int colscale_n;      // columns n < colscale_n are multiplied by colscale right after the bias (0: none)
  float colscale;
  bf16_t* save_dact;   // GELU'(pre) x dropout factor of the element, for the backward (see include/orbit2_hip.h)
  const bf16_t* mul;   // elementwise multiplier: the q14 factor tensor written through save_dact
  float rs_tile;       // (kernel-internal) the tile's row scale when the epilogue kind is 2
  float* colsum_ws;    // kind 3 only: fp32 [M / 256][N], row t = column sums of the bf16-rounded output over tile row t
};

__device__ __forceinline__ void epilogue4(const Epi& e, int m, int n, f32x4 v) {
  if (m >= e.M || n >= e.N) return;
  const size_t off = (size_t)m * e.ldc + n;
  if (e.bias) {
    const u32x2 b = *reinterpret_cast<const u32x2*>(e.bias + n);
    v[0] += bf2f((bf16_t)(b[0] & 0xffff)); v[1] += bf2f((bf16_t)(b[0] >> 16));
    v[2] += bf2f((bf16_t)(b[1] & 0xffff)); v[3] += bf2f((bf16_t)(b[1] >> 16));
  }
  if (n < e.colscale_n) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= e.colscale;
  }
  if (e.save_pre) {
    u32x2 o; o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(e.save_pre + off) = o;
    // the backward recomputes GELU'(pre) from the ROUNDED value: round here too so fwd == recompute
    v[0] = bf2f((bf16_t)(o[0] & 0xffff)); v[1] = bf2f((bf16_t)(o[0] >> 16));
    v[2] = bf2f((bf16_t)(o[1] & 0xffff)); v[3] = bf2f((bf16_t)(o[1] >> 16));
  }
  float dact[4] = {1.f, 1.f, 1.f, 1.f};
  if (e.save_dact && !e.save_pre) {                   // GELU of the bf16-rounded pre-activation, as with save_pre
    u32x2 o_; o_[0] = pack_bf2(v[0], v[1]); o_[1] = pack_bf2(v[2], v[3]);
    v[0] = bf2f((bf16_t)(o_[0] & 0xffff)); v[1] = bf2f((bf16_t)(o_[0] >> 16));
    v[2] = bf2f((bf16_t)(o_[1] & 0xffff)); v[3] = bf2f((bf16_t)(o_[1] >> 16));
  }
  if (e.act == 1) {
    if (e.save_dact) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { float g_; gelu_both(v[j], g_, dact[j]); v[j] = g_; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = gelu_fast(v[j]);
    }
  } else if (e.act == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
  }
  float r4[4] = {0.f, 0.f, 0.f, 0.f};
  if (e.residual) {
    const int rm = e.res_mod > 0 ? (m % e.res_mod) : m;
    const u32x2 rr = *reinterpret_cast<const u32x2*>(e.residual + (size_t)rm * e.ldr + n);
    r4[0] = bf2f((bf16_t)(rr[0] & 0xffff)); r4[1] = bf2f((bf16_t)(rr[0] >> 16));
    r4[2] = bf2f((bf16_t)(rr[1] & 0xffff)); r4[3] = bf2f((bf16_t)(rr[1] >> 16));
    if (e.res_first) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += r4[j];
    }
  }
  if (e.thr) {
    const uint64_t idx = ((uint64_t)m * (uint64_t)e.N + (uint64_t)n) >> 2;
    const uint32_t h = o2_hash64(e.seed ^ o2_seed_salt, idx);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool keep = ((h >> (8 * j)) & 0xffu) >= e.thr;
      v[j] = keep ? v[j] * e.dscale : 0.f;
      dact[j] = keep ? dact[j] * e.dscale : 0.f;
    }
  }
  if (e.save_dact) {
    u32x2 o; o[0] = pack_q14(dact[0], dact[1]); o[1] = pack_q14(dact[2], dact[3]);
    *reinterpret_cast<u32x2*>(e.save_dact + off) = o;
  }
  if (e.mul) {
    const u32x2 mu = *reinterpret_cast<const u32x2*>(e.mul + off);
    float m4[4];
    unpack_q14(mu[0], m4[0], m4[1]); unpack_q14(mu[1], m4[2], m4[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= m4[j];
  }
  if (e.dgelu_pre) {
    const u32x2 pr = *reinterpret_cast<const u32x2*>(e.dgelu_pre + off);
    v[0] *= dgelu_fast(bf2f((bf16_t)(pr[0] & 0xffff))); v[1] *= dgelu_fast(bf2f((bf16_t)(pr[0] >> 16)));
    v[2] *= dgelu_fast(bf2f((bf16_t)(pr[1] & 0xffff))); v[3] *= dgelu_fast(bf2f((bf16_t)(pr[1] >> 16)));
  }
  if (e.rowscale) {
    const float s = e.rowscale[m / e.rows_per_scale];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= s;
  }
  if (e.residual && !e.res_first) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += r4[j];
  }
  if (e.out_fp32) {
    float* c = reinterpret_cast<float*>(e.C) + off;
    if (e.beta != 0.f) {
      const f32x4 old = *reinterpret_cast<const f32x4*>(c);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += e.beta * old[j];
    }
    *reinterpret_cast<f32x4*>(c) = v;
  } else {
    bf16_t* c = reinterpret_cast<bf16_t*>(e.C) + off;
    if (e.beta != 0.f) {
      const u32x2 old = *reinterpret_cast<const u32x2*>(c);
      v[0] += e.beta * bf2f((bf16_t)(old[0] & 0xffff)); v[1] += e.beta * bf2f((bf16_t)(old[0] >> 16));
      v[2] += e.beta * bf2f((bf16_t)(old[1] & 0xffff)); v[3] += e.beta * bf2f((bf16_t)(old[1] >> 16));
    }
    u32x2 o; o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(c) = o;
  }
}

// ---- wide epilogue (bf16 output): one lane owns 8 consecutive n of one row m -------------------------------------
// The 16x16x32 accumulators give each lane 4 consecutive n per fragment; v_permlane16_swap between the fragments j
// and j+1 regroups them so that a lane holds 8 consecutive n (16 bytes of bf16): every epilogue load and store is a
// dwordx4 instead of two dwordx2 (the store tail is issue-bound), and loads are issued in batches ahead of the
// stores so no load queues behind a store (vmcnt retires in order).
__device__ __forceinline__ void unpack8(const u32x4& r, float* f) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = bf2f((bf16_t)(r[k] & 0xffff)); f[2 * k + 1] = bf2f((bf16_t)(r[k] >> 16)); }
}
__device__ __forceinline__ u32x4 pack8(const float* v) {
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
  return o;
}
struct Pre8 { u32x4 res, pre, old; };

__device__ __forceinline__ void epi8_load(const Epi& e, int m, int n, Pre8& q) {
  if (m >= e.M || n >= e.N) return;
  const size_t off = (size_t)m * e.ldc + n;
  if (e.residual) {
    const int rm = e.res_mod > 0 ? (m % e.res_mod) : m;
    q.res = *reinterpret_cast<const u32x4*>(e.residual + (size_t)rm * e.ldr + n);
  }
  if (e.dgelu_pre) q.pre = *reinterpret_cast<const u32x4*>(e.dgelu_pre + off);
  else if (e.mul) q.pre = *reinterpret_cast<const u32x4*>(e.mul + off);     // (never both: gemm_make_epi)
  if (e.beta != 0.f) q.old = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(e.C) + off);
}

// EK: 0 = every option of Epi is a runtime test (uniform branches: fine with two waves per SIMD, a serial chain with one);
// 1 / 2 / 3 = the three hot combinations of the Block on whole tiles, decided by the host (w4_epi_kind) and folded at compile
// time so that the 4-wave kernel's epilogue is straight-line code the compiler can interleave across rows:
//   1: bias + GELU + saved GELU' factor + dropout (fc1 forward)      2: bias + dropout + row scale (one per tile; optional) + residual
//   3: x saved factor (fc2 input gradient)                              (proj / fc2 forward)
// Same expressions in the same order as EK = 0: the bits do not depend on the path.
#define O2_OPT(on_kinds, runtime) (EK == 0 ? (runtime) : (on_kinds))
template <int EK = 0>
__device__ __forceinline__ void epi8_finish(const Epi& e, int m, int n, float* v, const float* bias8, const Pre8& q) {
  if (EK == 0 && (m >= e.M || n >= e.N)) return;
  const size_t off = (size_t)m * e.ldc + n;
  if (O2_OPT(EK == 1 || EK == 2, e.bias != nullptr)) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += bias8[k];
  }
  if (O2_OPT(false, n < e.colscale_n)) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= e.colscale;
  }
  if (O2_OPT(false, e.save_pre != nullptr)) {
    const u32x4 o = pack8(v);
    *reinterpret_cast<u32x4*>(e.save_pre + off) = o;
    unpack8(o, v);   // the backward recomputes GELU'(pre) from the ROUNDED value: round here too
  }
  float dact[8];
  // (GELU of the bf16-ROUNDED pre-activation, as with save_pre: the forward's bits do not depend on which of the two the
  // caller asked for)
  if (O2_OPT(EK == 1, e.save_dact && !e.save_pre)) { const u32x4 o_ = pack8(v); unpack8(o_, v); }
  if (O2_OPT(EK == 1, e.act == 1)) {
    if (O2_OPT(EK == 1, e.save_dact != nullptr)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) gelu_both(v[k], v[k], dact[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = gelu_fast(v[k]);
    }
  } else if (O2_OPT(false, e.act == 2)) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
  }
  float r8[8];
  if (O2_OPT(EK == 2, e.residual != nullptr)) {
    unpack8(q.res, r8);
    if (O2_OPT(false, e.res_first != 0)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += r8[k];
    }
  }
  if (O2_OPT(EK == 1 || EK == 2, e.thr != 0)) {
    const uint64_t idx = ((uint64_t)m * (uint64_t)e.N + (uint64_t)n) >> 2;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const uint32_t h = o2_hash64(e.seed ^ o2_seed_salt, idx + hlf);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool keep = ((h >> (8 * j)) & 0xffu) >= e.thr;
        v[4 * hlf + j] = keep ? v[4 * hlf + j] * e.dscale : 0.f;
        if (O2_OPT(EK == 1, e.save_dact != nullptr)) dact[4 * hlf + j] = keep ? dact[4 * hlf + j] * e.dscale : 0.f;
      }
    }
  }
  if (O2_OPT(EK == 1, e.save_dact != nullptr)) {
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack_q14(dact[2 * k], dact[2 * k + 1]);
    *reinterpret_cast<u32x4*>(e.save_dact + off) = o;
  }
  if (O2_OPT(EK == 3, e.mul != nullptr)) {
    float m8[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) unpack_q14(q.pre[k], m8[2 * k], m8[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= m8[k];
  }
  if (O2_OPT(false, e.dgelu_pre != nullptr)) {
    float p8[8];
    unpack8(q.pre, p8);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= dgelu_fast(p8[k]);
  }
  if (O2_OPT(EK == 2, e.rowscale != nullptr)) {
    const float s = EK == 2 ? e.rs_tile : e.rowscale[m / e.rows_per_scale];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= s;
  }
  if (O2_OPT(EK == 2, e.residual && !e.res_first)) {
#pragma clang fp contract(off)   // with the options folded (EK == 2) the row scale's multiply is adjacent: no fused multiply-add here,
                                 // the runtime form has none (the two lie in different blocks) and the bits must not depend on the path
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += r8[k];
  }
  if (O2_OPT(false, e.beta != 0.f)) {
    float o8[8];
    unpack8(q.old, o8);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += e.beta * o8[k];
  }
  *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(e.C) + off) = pack8(v);
}

// XCD-aware remap (bijective for any grid size): blocks b, b+8, .. share an XCD -> give each XCD a
// contiguous range of tile ids.
__device__ __forceinline__ int xcd_tile_id() {
  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
}

// Round-major form for the kernels that hold ONE workgroup per CU (the 4-wave 256 x 256 kernel): the 256 workgroups that run at
// the same time (block b lands on XCD b & 7, in dispatch order) take 256 CONSECUTIVE tile ids -- a compact block of the output --
// and each XCD 32 consecutive ids of those (its L2 cohort, 4 x 8 tiles).  With xcd_tile_id the eight XCDs work on eight distant
// id ranges: a panel strip that two of them need is needed milliseconds apart and comes from HBM twice; here the 8 cohorts of a
// round sweep K side by side and the second reader of a strip finds it in the Infinity Cache.  What that buys: the clock
// (profiles/r06_dw_traffic_clock.txt, r06_dw_traffic_instep.txt: the weight-gradient launch with its panels served from the
// Infinity Cache runs 13 % faster at the same MFMA-busy share).  Bijective for any grid size.
__device__ __forceinline__ int xcd_round_tile_id() {
  const int nwg = gridDim.x, b = blockIdx.x;
  const int base = b & ~255;
  const int cnt = (nwg - base) < 256 ? (nwg - base) : 256;          // workgroups of this round
  const int x = b & 7, s = (b & 255) >> 3;
  const int q8 = cnt >> 3, r8 = cnt & 7;
  return base + (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + s;
}
// Start barrier of a round's XCD cohort (long contractions only: the weight-gradient launch, 2048 K-tiles per tile).  Within a
// round the 32 workgroups of an XCD keep step by themselves -- same instruction stream, same clock -- and fetch every strip
// K-tile ONCE beyond L2 (profiles/r06_l2_share_probe.txt: one round = 1.00-1.01 x the ideal bytes at any K).  What breaks the
// sharing is the round change: the next round's workgroups start as CUs fall free, microseconds apart, L2 holds ~10 K-tiles
// (12 us) of a cohort's 12 strips, and a workgroup that starts outside that window re-fetches its strips for the whole sweep
// (three rounds: 1.19 x; the Block's grouped launch, 6.75 rounds: 1.7 x = 80 GB for 48).  So a workgroup waits at its tile's
// start until its cohort (round b >> 8, XCD b & 7) has assembled: a sense-reversing counter pair per cohort in a static device
// array, self-cleaning (the last arriver zeroes the count and bumps the generation), polled by one lane with s_sleep between
// agent-scope loads, BOUNDED (~15 us): it is a pacing hint, never needed for correctness -- a cohort that does not assemble
// (another stream's kernel holding CUs) costs its workgroups the bound once per 2.7 ms tile.
#define O2_W4_SYNC_SLOTS 512
static __device__ unsigned int o2_w4_sync[3 * O2_W4_SYNC_SLOTS];       // [slot] = arrivals, [SLOTS + slot] = generation,
                                                                        // [2 SLOTS + slot] = arrivals at the in-sweep check points
__device__ __forceinline__ const unsigned int* w4_cohort_start(int& n_out) {
  const int nwg = gridDim.x, b = blockIdx.x;
  const int base = b & ~255, x = b & 7;
  const int cnt = (nwg - base) < 256 ? (nwg - base) : 256;
  const int n = (cnt >> 3) + (x < (cnt & 7) ? 1 : 0);               // workgroups of this round on this XCD
  const int slot = ((b >> 8) << 3) + x;
  if (slot < O2_W4_SYNC_SLOTS && n > 1) {
    if (threadIdx.x == 0) {
      unsigned int* cntp = o2_w4_sync + slot;
      unsigned int* genp = o2_w4_sync + O2_W4_SYNC_SLOTS + slot;
      const unsigned int g0 = __hip_atomic_load(genp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int old = __hip_atomic_fetch_add(cntp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the FIRST arriver zeroes the sweep's check-point counter: nobody of this cohort can reach a check point (64 K-tiles =
      // ~85 us into its sweep) before that, whereas the LAST arriver may come after a member that gave up waiting has already
      // counted itself there
      if (old == 0u) __hip_atomic_store(o2_w4_sync + 2 * O2_W4_SYNC_SLOTS + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1u >= (unsigned int)n) {
        __hip_atomic_store(cntp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(genp, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        for (int polls = 0; polls < 56; ++polls) {
          if (__hip_atomic_load(genp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g0) break;
          __builtin_amdgcn_s_sleep(8);
        }
      }
    }
    __syncthreads();
    n_out = n;
    return o2_w4_sync + 2 * O2_W4_SYNC_SLOTS + slot;
  }
  n_out = 0;
  return nullptr;
}

#ifndef O2_W4_WALK
#define O2_W4_WALK 1        // 1: round-major ids for the 4-wave kernels; 0: the contiguous-range-per-XCD ids of rounds 2-5 (A/B builds)
#endif
__device__ __forceinline__ int w4_tile_id() { return O2_W4_WALK ? xcd_round_tile_id() : xcd_tile_id(); }

// one 128x128 output tile (tile `id` of the problem; tiles are walked in groups of 8 tile-rows so neighbours
// share panels)
// MT = rows of the tile: 128, or 64 (A K-contiguous only) -- half the rows per workgroup, twice the workgroups: problems whose
// 128 x 128 tiles number fewer than two per CU (the N = 1024 GEMMs of interm_117m: 256 tiles) then still put TWO workgroups on
// every CU, and a lone workgroup's exposed waits (fragment reads, the barrier, the LDS-DMA pieces) hide behind its partner's MFMAs
template <bool A_KC, bool B_KC, int MT = 128>
__device__ __forceinline__ void gemm128_tile(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M, int N,
                                             int K, int lda, int ldb, int tiles_m, int tiles_n, const Epi& epi,
                                             int id, char* smem) {
  static_assert(MT == 128 || (MT == 64 && A_KC), "64-row tiles: K-contiguous A");
  constexpr int FI = MT / 32;            // m-side fragments per wave: 4 (64 rows of 128) or 2 (32 rows of 64)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int GROUP = 8 * (128 / MT);
  const int per_group = GROUP * tiles_n;
  const int grp = id / per_group;
  const int first_m = grp * GROUP;
  const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (id % per_group) % gsz;
  const int tn = (id % per_group) / gsz;
  const int m0 = tm * MT, n0 = tn * BN;

  f32x4 acc[FI][4];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;      // a ragged K tail is staged from the zero page
  stage_tile<A_KC, MT / 32>(A, lda, m0, M, 0, K, smem, wave, lane);
  stage_tile<B_KC>(B, ldb, n0, N, 0, K, smem + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    char* sa = smem + cur * 2 * TILE_BYTES;
    char* sb = sa + TILE_BYTES;
    if (kt + 1 < nk) {
      char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile<A_KC, MT / 32>(A, lda, m0, M, (kt + 1) * BK, K, na, wave, lane);
      stage_tile<B_KC>(B, ldb, n0, N, (kt + 1) * BK, K, na + TILE_BYTES, wave, lane);
    }
    // both k-halves' fragments are read up front (the second half lands under the first half's MFMAs), and the
    // MFMA bursts run at raised priority so the co-resident workgroup's loads do not break them up
    bf16x8 fa[2][FI], fb[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < FI; ++i) fa[kk][i] = read_frag<A_KC>(sa, wm * (MT / 2) + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[kk][j] = read_frag<B_KC>(sb, wn * 64 + j * 16, kk, lane);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // D[i = n][j = m]: lane holds m = lane&15, n = 4*(lane>>4) + r
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    const int m = m0 + wm * (MT / 2) + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      epilogue4(epi, m, n, acc[i][j]);
    }
  }
}

template <bool A_KC, bool B_KC, int MT = 128>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         int M, int N, int K, int lda, int ldb, int tiles_m,
                                                         int tiles_n, Epi epi) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [stage][A|B]
  gemm128_tile<A_KC, B_KC, MT>(A, B, M, N, K, lda, ldb, tiles_m, tiles_n, epi, xcd_tile_id(), smem);
}

// Grouped launch: up to O2_GEMM_MAX_GROUP independent problems of one operand form share a grid, so the tail
// round of each (tiles mod 512 workgroup slots) is filled by the next problem's tiles.  Used for the four weight
// gradients of a transformer block: 576 + 1728 + 2304 + 2304 tiles in 14 rounds instead of 2 + 4 + 5 + 5.
struct GProb {
  const bf16_t* A;
  const bf16_t* B;
  int M, N, K, lda, ldb, tiles_m, tiles_n, tile_end;
  Epi epi;
};
struct GArgs {
  int n;
  int pace;        // 4-wave kernel: cohort start barrier (w4_cohort_start) -- long contractions, round-major ids
  GProb p[ORBIT2_GEMM_MAX_GROUP];
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm128_grouped_kernel(GArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  const int id = xcd_tile_id();
  int pi = 0;
  while (pi + 1 < g.n && id >= g.p[pi].tile_end) ++pi;
  const int first = pi ? g.p[pi - 1].tile_end : 0;
  const GProb& P = g.p[pi];
  const Epi epi = P.epi;
  gemm128_tile<A_KC, B_KC>(P.A, P.B, P.M, P.N, P.K, P.lda, P.ldb, P.tiles_m, P.tiles_n, epi, id - first, smem);
}

// epilogue of a 256x256 tile held as 8 waves x (8 x 4) 16x16 accumulators.  Wave (wm, wn) owns the rows
// wm*MW + (i>>2)*MH + (i&3)*16 (+16) of fragment row i and the columns wn*NW + (j>>1)*NH + (j&1)*16 (+16) of fragment column j:
//   MW 128, MH 64, NW 64, NH 32 : a contiguous 128 x 64 wave tile (ring kernel, 8-phase NT kernel)
//   MW 64, MH 128, NW 32, NH 128: two 64-row / 32-column groups half a tile apart (8-phase kernel with contiguous units)
template <int MW = 128, int MH = 64, int NW = 64, int NH = 32>
__device__ __forceinline__ void epilogue256(const Epi& epi, f32x4 (&acc)[8][4], int m0, int n0, int wm, int wn, int lane) {
  if (epi.out_fp32) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wm * MW + (i >> 2) * MH + (i & 3) * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * NW + (j >> 1) * NH + (j & 1) * 16 + 4 * (lane >> 4);
        epilogue4(epi, m, n, acc[i][j]);
      }
    }
    return;
  }
  // wide path: after the swap of fragments (2jp, 2jp+1), the lane in 16-lane row r4 owns
  // n = nw + NH*jp + 16*(r4&1) + 8*(r4>>1) + {0..7} of row m
  const int r4 = lane >> 4;
  const int nw = n0 + wn * NW + 16 * (r4 & 1) + 8 * (r4 >> 1);
  float bias8[2][8];
  if (epi.bias) {
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const int n = nw + NH * jp;
      u32x4 b = {0u, 0u, 0u, 0u};
      if (n < epi.N) b = *reinterpret_cast<const u32x4*>(epi.bias + n);
      unpack8(b, bias8[jp]);
    }
  }
#pragma unroll
  for (int ib = 0; ib < 8; ib += 2) {
    Pre8 q[2][2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
        epi8_load(epi, m0 + wm * MW + ((ib + ii) >> 2) * MH + ((ib + ii) & 3) * 16 + (lane & 15), nw + NH * jp, q[ii][jp]);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int m = m0 + wm * MW + ((ib + ii) >> 2) * MH + ((ib + ii) & 3) * 16 + (lane & 15);
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[ib + ii][2 * jp][c]),
                                                           __float_as_uint(acc[ib + ii][2 * jp + 1][c]), false, false);
          v[c] = __uint_as_float(sw[0]);
          v[4 + c] = __uint_as_float(sw[1]);
        }
        epi8_finish(epi, m, nw + NH * jp, v, bias8[jp], q[ii][jp]);
      }
    }
  }
}

#ifdef O2_STAMP
// Diagnostic build only (tools/stamp_build.sh, never the shipped library): waves 0 and 4 of the first 64 workgroups sum
// the shader cycles they spend in each segment of the main loop; nothing in the kernel reads these words.
__device__ unsigned int o2_dbg[64 * 2 * 8];
#define O2_T() ((unsigned)__builtin_amdgcn_s_memtime())
#endif
constexpr int BM2 = 256, BN2 = 256;
#ifdef O2_STAMP
#define O2_SEG(acc) { t1 = O2_T(); acc += t1 - t0; t0 = t1; }
#else
#define O2_SEG(acc)
#endif

// ==========================================================================================
// 256x256x64 tile, 512 threads = 8 waves (2 x 4), every operand form (NT forward GEMMs, NN input gradients on the stored
// weight, TN weight gradients).  Eight phases per two K-tiles: a phase = one 64 x 32 quadrant of a wave's 128 x 64
// accumulators over a 64-deep K-tile = 16 MFMAs, fed by 0 / 4 / 8 / 12 fragment reads, with ONE 16-KiB "unit" of a later
// K-tile put in flight by LDS-DMA (two 1-KiB pieces per wave).
//   * 64-deep K-tiles: every LDS-DMA lane group fetches whole 128-byte lines (the 32-deep ring kernel of round 1 fetched
//     half lines: twice the requests for the same bytes; its stamps -- profiles/r02_gemm_stamps.txt -- also showed the two
//     LDS-DMA pieces it issued between the MFMAs stretching every 512-cycle MFMA segment to ~685).
//   * unit h (h = 0, 1) of an operand = tile rows / columns h*128 + [0, 128), 64 k deep.  K-contiguous operand: image
//     [128 rows][64 k] (128-byte rows, 16-byte chunk c of unit row u stored at c ^ (u & 7): conflict-free ds_read_b128).
//     K-strided operand (fetched as rows of k, so a unit must be 128 CONTIGUOUS columns = whole 256-byte pieces): image
//     [64 k][128 cols] with the 128-tile kernel's granule swizzle, fragments by ds_read_b64_tr_b16.
//     Wave (wm, wn) owns rows h*128 + wm*64 + [0, 64) and columns h*128 + wn*32 + [0, 32) of BOTH halves h, so every wave
//     consumes a unit in the same phase (epilogue256<64, 128, 32, 128>).  LDS = 2 K-tiles x 4 units = 128 KiB; a unit is
//     re-filled as soon as its last reader has passed.
//   * schedule of K-tile t (buffer t & 1); fragments of B-h0 stay in registers for p3:
//       p0: issue A-h1(t+1) | read B-h0, A-h0 | quadrant (0,0)
//       p1: issue B-h0(t+2) | read B-h1       | quadrant (0,1)
//       p2: issue A-h0(t+2) | read A-h1       | quadrant (1,1)
//       p3: issue B-h1(t+2) | vmcnt(6)        | quadrant (1,0)
//     Every phase is  L: issue + reads | s_barrier | lgkmcnt(0), M: 16 MFMAs | s_barrier ; waves 4-7 run one barrier behind
//     waves 0-3, so on every SIMD one wave is in M while its partner is in L.  Fragment reads stay IN FLIGHT across the
//     barrier that ends L (their latency hides behind the barrier's).
//   * hazards.  RAW: the vmcnt(6) in L(t,p3) leaves the three youngest units (issued in p1..p3: K-tile t+2's) in flight, so
//     every unit of K-tile t+1 (the youngest of them issued in p0) has landed for this wave; both wave groups pass that
//     wait before the barrier that ends M(t,p3) of waves 0-3, and the first read of K-tile t+1 is in L(t+1,p0), after it.
//     WAR: with reads in flight across a barrier the other wave group retires its reads one barrier later, so a unit may
//     only be re-filled TWO phases after its last read: A-h1 (read p2 / issued p0 of the next K-tile), A-h0 (p0 / p2),
//     B-h1 (p1 / p3) keep that distance; B-h0 (read p0, re-filled p1) does not, so p0 issues its B reads FIRST and retires
//     them -- only them -- with a counted lgkmcnt before its barrier (LDS reads return in order).
//   K-tiles past the end re-load the last one into a slot nobody reads again: branch-free loop, constant vmcnt arithmetic.
//   Results are bit-identical to the 128-tile kernel's (same k order per accumulator).
// ==========================================================================================
constexpr int BK3 = 64;
constexpr int UNIT3 = 128 * 64 * 2;   // 16 KiB

__device__ __forceinline__ bf16x8 read_frag3(const char* unit, int ubase, int kk, int lane) {
  const int u = ubase + (lane & 15);
  const int c = kk * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(unit + u * 128 + ((c ^ (u & 7)) << 4));
}

template <bool KC>
__device__ __forceinline__ void stage_unit4(const bf16_t* __restrict__ G, int ld, int r0, int rmax, int k0, int h,
                                            char* unit, int wave, int lane) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = wave * 2 + t;               // 1-KiB piece 0..15 of the unit
    if (KC) {
      const int u = i * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (u & 7);
      int gr = r0 + h * 128 + u;
      gr = gr < rmax ? gr : rmax - 1;
      glds16(G + (size_t)gr * ld + k0 + chunk * 8, unit + i * 1024);
    } else {
      const int krow = i * 4 + (lane >> 4);
      const int cp = lane & 15;
      const int sw = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int chunk = ((((cp >> 1) ^ sw)) << 1) | (cp & 1);
      int col = r0 + h * 128 + chunk * 8;
      col = col <= rmax - 8 ? col : rmax - 8;
      glds16(G + (size_t)(k0 + krow) * ld + col, unit + i * 1024);
    }
  }
}

template <bool KC>
__device__ __forceinline__ bf16x8 read_frag4(const char* unit, int ubase, int kk, int lane) {
  if (KC) return read_frag3(unit, ubase, kk, lane);
  return read_frag<false>(unit, ubase, kk, lane);
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm256t_tile(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M, int N,
                                              int K, int lda, int ldb, int tiles_m, int tiles_n, const Epi& epi, int id,
                                              char* smem) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  constexpr int GROUP = 4;                       // an XCD's 32 concurrent tiles: 4 tile rows x 8 tile columns
  const int per_group = GROUP * tiles_n;
  const int grp = id / per_group;
  const int first_m = grp * GROUP;
  const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (id % per_group) % gsz;
  const int tn = (id % per_group) / gsz;
  const int m0 = tm * BM2, n0 = tn * BN2;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK3;
  auto issue = [&](int t, int which) {           // which: 0 = A half 0, 1 = A half 1, 2 = B half 0, 3 = B half 1
    const int kt = t < nk ? t : nk - 1;
    char* unit = smem + ((t & 1) * 4 + which) * UNIT3;
    if (which < 2) stage_unit4<A_KC>(A, lda, m0, M, kt * BK3, which, unit, wave, lane);
    else stage_unit4<B_KC>(B, ldb, n0, N, kt * BK3, which - 2, unit, wave, lane);
  };
  issue(0, 0); issue(0, 2); issue(0, 3); issue(0, 1);
  issue(1, 2); issue(1, 0); issue(1, 3);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wave >= 4) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind
#ifdef O2_STAMP
  unsigned tL = 0, tBa = 0, tM = 0, tV = 0, tBb = 0, t0 = O2_T(), t1;
  const unsigned tstart = t0;
#endif
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define O2_MFMA_Q(MI, NJ, FB)                                                                              \
  __builtin_amdgcn_s_setprio(1);                                                                           \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                         \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
    acc[(MI) + i][(NJ) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j][kk], fa[i][kk], acc[(MI) + i][(NJ) + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);
  for (int t = 0; t < nk; ++t) {
    const char* ub = smem + (t & 1) * 4 * UNIT3;
    // ---- p0
    issue(t + 1, 1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fb0[j][kk] = read_frag4<B_KC>(ub + 2 * UNIT3, wn * 32 + j * 16, kk, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[i][kk] = read_frag4<A_KC>(ub, wm * 64 + i * 16, kk, lane);
    __builtin_amdgcn_sched_barrier(0);
    // LDS reads return in order: all but the A reads (8 ds_read_b128, or 16 transposing reads -- 15 is the counter's
    // largest encodable value) have landed = every B-nh0 read, the unit that p1 re-fills
    __builtin_amdgcn_s_waitcnt(A_KC ? 0xC87F : 0xCF7F);
    O2_SEG(tL)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    O2_SEG(tBa)
    O2_MFMA_Q(0, 0, fb0)
    O2_SEG(tM)
    __builtin_amdgcn_s_barrier();
    O2_SEG(tBb)
    // ---- p1
    issue(t + 2, 2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fb1[j][kk] = read_frag4<B_KC>(ub + 3 * UNIT3, wn * 32 + j * 16, kk, lane);
    __builtin_amdgcn_sched_barrier(0);
    O2_SEG(tL)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    O2_SEG(tBa)
    O2_MFMA_Q(0, 2, fb1)
    O2_SEG(tM)
    __builtin_amdgcn_s_barrier();
    O2_SEG(tBb)
    // ---- p2
    issue(t + 2, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[i][kk] = read_frag4<A_KC>(ub + UNIT3, wm * 64 + i * 16, kk, lane);
    __builtin_amdgcn_sched_barrier(0);
    O2_SEG(tL)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    O2_SEG(tBa)
    O2_MFMA_Q(4, 2, fb1)
    O2_SEG(tM)
    __builtin_amdgcn_s_barrier();
    O2_SEG(tBb)
    // ---- p3
    issue(t + 2, 3);
    O2_SEG(tL)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    O2_SEG(tV)
    __builtin_amdgcn_s_barrier();
    O2_SEG(tBa)
    O2_MFMA_Q(4, 0, fb0)
    O2_SEG(tM)
    __builtin_amdgcn_s_barrier();
    O2_SEG(tBb)
  }
#undef O2_MFMA_Q
  if (wave < 4) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dead-slot loads before LDS is released
#ifdef O2_STAMP
  if (blockIdx.x < 64 && (wave & 3) == 0 && lane == 0) {
    unsigned* d = o2_dbg + (blockIdx.x * 2 + (wave >> 2)) * 8;
    d[0] = tL; d[1] = tBa; d[2] = tM; d[3] = tV; d[4] = tBb; d[5] = O2_T() - tstart; d[6] = (unsigned)nk; d[7] = 2;
  }
#endif
  epilogue256<64, 128, 32, 128>(epi, acc, m0, n0, wm, wn, lane);
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 2) void gemm256t_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                          int M, int N, int K, int lda, int ldb, int tiles_m,
                                                          int tiles_n, Epi epi) {
  __shared__ __attribute__((aligned(16))) char smem[8 * UNIT3];
  gemm256t_tile<A_KC, B_KC>(A, B, M, N, K, lda, ldb, tiles_m, tiles_n, epi, xcd_tile_id(), smem);
}

// grouped form (the four weight gradients of a Block in one grid: 144 + 432 + 576 + 576 tiles fill 6.75 rounds of the 256
// CUs instead of 0.56 + 1.69 + 2.25 + 2.25 one by one)
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 2) void gemm256t_grouped_kernel(GArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[8 * UNIT3];
  const int id = xcd_tile_id();
  int pi = 0;
  while (pi + 1 < g.n && id >= g.p[pi].tile_end) ++pi;
  const int first = pi ? g.p[pi - 1].tile_end : 0;
  const GProb& P = g.p[pi];
  const Epi epi = P.epi;
  gemm256t_tile<A_KC, B_KC>(P.A, P.B, P.M, P.N, P.K, P.lda, P.ldb, P.tiles_m, P.tiles_n, epi, id - first, smem);
}

// ==========================================================================================
// 256x256x64 tile, 256 threads = 4 waves, ONE wave per SIMD with the whole register file: a wave owns a 128 x 128 quadrant
// (8 x 8 blocks of v_mfma_f32_16x16x32_bf16 = 256 accumulator registers a[0:255]), 128 MFMAs per K-tile fed by 32
// ds_read_b128 (0.25 reads per MFMA against 0.375 in the 8-wave kernel) and 16 LDS-DMA pieces per wave.  The main loop is
// one asm statement whose instruction stream -- every MFMA, LDS read, LDS-DMA piece, counted wait and barrier assigned to
// its place -- is generated by tools/gen_gemm_w4.py (csrc/gemm_w4_asm.h; the schedule is documented there).  LDS image,
// swizzle and tile walk are the 8-phase kernel's (units of 128 rows x 64 k, 2 stages x 4 units = 128 KiB).
// Whole tiles only (M, N % 256 == 0, K % 64 == 0): the LDS-DMA addresses are buffer offsets (one per-lane voffset per
// operand, a scalar offset per piece), rows are not clamped.
// ==========================================================================================
#include "gemm_w4_asm.h"

#ifdef O2_W4_STAMP
__device__ unsigned int o2_dbg_w4[64 * 4 * 16];   // diagnostic build only: per-wave cycle sums of the loop's segments
extern "C" int orbit2_debug_read_w4(unsigned int* host_dst, int n) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(o2_dbg_w4), sizeof(unsigned) * (size_t)n);
}
#endif
#ifdef O2_W4_TRACE
// diagnostic build only (tools/w4_trace.py): when did each workgroup start its K sweep and when did it end it -- s_memrealtime
// (100 MHz) of wave 0 at the statement's entry and exit, [blockIdx][2]: the start spread of a round's XCD cohort and the drift
// between its workgroups over a sweep
__device__ unsigned long long o2_w4_trace[32768 * 2];
extern "C" int orbit2_debug_read_w4_trace(unsigned long long* host_dst, int n) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(o2_w4_trace), sizeof(unsigned long long) * (size_t)n);
}
#define O2_W4_TRACE_POINT(k) if (tid == 0 && blockIdx.x < 32768) o2_w4_trace[blockIdx.x * 2 + (k)] = __builtin_amdgcn_s_memrealtime()
#else
#define O2_W4_TRACE_POINT(k)
#endif
// Epilogue of the 4-wave kernel: the accumulators go through LDS (free once the main loop is over) as an fp32 image of
// 128 tile rows x 256 columns (two passes: accumulator rows i = 0..3, then 4..7 of every wave), written straight from the
// accumulator registers (ds_write_b128 a[..], 16-byte chunk c of image row r stored at c ^ (r & 7): conflict-free writes and
// reads), and leave it row by row: a lane owns 8 consecutive n of a row, 32 lanes one 512-byte row segment, so every
// epilogue load and store is a whole-line dwordx4 -- and the epilogue code (every runtime option of Epi) exists ONCE, in
// a loop, instead of once per accumulator block (the unrolled per-block form of the 8-wave kernel is 100+ KB of code).
__device__ __forceinline__ void w4_epilogue_rows(const Epi& epi, const char* smem, int m0, int n0, int hh, int tid) {
  const int q = tid & 31;
  const int n = n0 + 8 * q;
  float bias8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias8[k] = 0.f;
  if (epi.bias && !epi.out_fp32) {
    u32x4 b = {0u, 0u, 0u, 0u};
    if (n < epi.N) b = *reinterpret_cast<const u32x4*>(epi.bias + n);
    unpack8(b, bias8);
  }
  // lean path (bias / column scale only; whole tiles by construction of this kernel): per 8 values two LDS reads, the adds
  // and multiplies of epi8_finish in its order (bit-identical), four conversions and one 16-byte store
  const bool lean = !epi.out_fp32 && !epi.save_pre && epi.act == 0 && !epi.residual && epi.thr == 0 && !epi.dgelu_pre &&
                    !epi.rowscale && epi.beta == 0.f && !epi.mul;
  if (lean) {
    const float mul = n < epi.colscale_n ? epi.colscale : 1.0f;
    const int r8 = tid >> 5;
    bf16_t* cp = reinterpret_cast<bf16_t*>(epi.C) + (size_t)(m0 + hh * 64 + r8) * epi.ldc + n;
    const size_t step8 = (size_t)8 * epi.ldc;
    const char* lp = smem + r8 * 1024;
    const uint32_t c0 = (uint32_t)(((2 * q) ^ r8) << 4), c1 = (uint32_t)(((2 * q + 1) ^ r8) << 4);   // (it*8 + r8) & 7 == r8
#pragma unroll      // all 16 rows in flight: +0.6 % on the K = 3072 GEMMs over unroll 4 (profiles/r04_gemm_lean_unroll.txt), same bits
    for (int it = 0; it < 16; ++it) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(lp + it * 8192 + c0);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(lp + it * 8192 + c1);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (v[k] + bias8[k]) * mul;
      *reinterpret_cast<u32x4*>(cp + (size_t)(it + (it >= 8 ? 8 : 0)) * step8) = pack8(v);
    }
    return;
  }
  if (epi.out_fp32) {
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
      const int lr = it * 8 + (tid >> 5);                            // image row 0..127
      const int m = m0 + (lr >> 6) * 128 + hh * 64 + (lr & 63);
      const char* row = smem + lr * 1024;
      const int sw = lr & 7;
      epilogue4(epi, m, n, *reinterpret_cast<const f32x4*>(row + (((2 * q) ^ sw) << 4)));
      epilogue4(epi, m, n + 4, *reinterpret_cast<const f32x4*>(row + (((2 * q + 1) ^ sw) << 4)));
    }
    return;
  }
}

// generic bf16 epilogue of the 4-wave kernel (any combination of Epi's options): two image rows per step and lane, the
// loads (residual / pre-activation / old C) two steps ahead of their use -- the first two steps' loads are issued BEFORE the
// accumulators are staged through LDS: with one wave per SIMD nobody else hides their latency (at one step ahead the
// epilogue ran at the ~2 TB/s its 32 KB in flight per CU allow)
__device__ __forceinline__ int w4_row_m(int m0, int hh, int tid, int it, int k) {
  const int lr = (it * 2 + k) * 8 + (tid >> 5);
  return m0 + (lr >> 6) * 128 + hh * 64 + (lr & 63);
}
__device__ __forceinline__ void w4_generic_load(const Epi& epi, int m0, int n, int hh, int tid, int it, Pre8 (&pq)[2]) {
#pragma unroll
  for (int k = 0; k < 2; ++k) epi8_load(epi, w4_row_m(m0, hh, tid, it, k), n, pq[k]);
}
__device__ __forceinline__ void w4_generic_finish(const Epi& epi, const char* smem, int m0, int n, int hh, int tid, int it,
                                                  const float* bias8, const Pre8 (&pq)[2]) {
  const int q = tid & 31;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int lr = (it * 2 + k) * 8 + (tid >> 5);
    const char* row = smem + lr * 1024;
    const int sw = lr & 7;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(row + (((2 * q) ^ sw) << 4));
    const f32x4 hi = *reinterpret_cast<const f32x4*>(row + (((2 * q + 1) ^ sw) << 4));
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    epi8_finish(epi, w4_row_m(m0, hh, tid, it, k), n, v, bias8, pq[k]);
  }
}

// FORM: 0 = NT (A, B K-contiguous), 1 = NN (B K-strided: the stored weight in dX = dY.W), 2 = TN (both K-strided: dW = dY^T.X),
// 3 = TT.  STAMP (diagnostic build only): per-segment cycle sums of the loop into o2_dbg_w4.
template <int FORM, bool STAMP, int EK = 0>
__device__ __forceinline__ void gemm256w_tile(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M, int N,
                                              int K, int lda, int ldb, int tiles_m, int tiles_n, const Epi& epi, int id,
                                              char* smem, const unsigned int* pace_ctr = nullptr, int pace_n = 0) {
  constexpr bool A_KC = (FORM == 0 || FORM == 1), B_KC = (FORM == 0 || FORM == 3);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
#ifndef O2_W4_GROUP
#define O2_W4_GROUP 4
#endif
  constexpr int GROUP = O2_W4_GROUP;
  // ids walk groups of GROUP tile-rows, m fastest: 32 consecutive ids = 4 x 8 tiles (12 panel strips per K-tile for 32 tiles).  A
  // problem that is wider than tall (tiles_n > tiles_m: the fc2 weight gradient, 12 x 48) is walked the other way round, groups of
  // GROUP tile-columns with n fastest: the 256 consecutive ids of a round (xcd_round_tile_id) then cover ~21 x 12 tiles = 33 strips
  // instead of 5 x 48 = 53
#ifndef O2_W4_TWALK
#define O2_W4_TWALK 1
#endif
  const bool tw = O2_W4_TWALK && tiles_n > tiles_m;
  const int t_major = tw ? tiles_n : tiles_m, t_minor = tw ? tiles_m : tiles_n;
  const int per_group = GROUP * t_minor;
  const int grp = id / per_group;
  const int first_g = grp * GROUP;
  const int gsz = (t_major - first_g) < GROUP ? (t_major - first_g) : GROUP;
  // (measured and not kept, round 6: the minor index fastest inside 8-wide chunks -- four workgroups dispatched together then hold
  // four different minor strips: the same time, 11 % more bytes fetched; profiles/r06_nfast_ab.txt, r06_w4_trace_nfast.txt)
  const int t_a = first_g + (id % per_group) % gsz;
  const int t_b = (id % per_group) / gsz;
  const int tm = tw ? t_b : t_a, tn = tw ? t_a : t_b;
  const int m0 = tm * BM2, n0 = tn * BN2;

#ifdef O2_W4_STAMP
  const unsigned ts0 = (unsigned)__builtin_amdgcn_s_memtime();
  unsigned ts[8];
#define O2_TS(k) ts[k] = (unsigned)__builtin_amdgcn_s_memtime()
#else
#define O2_TS(k)
#endif
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int u = lane & 15, cg = lane >> 4, sw = lane & 7;
  // fragment read bases (stage 0) and their stage flips t = 2 base + stage size (base <- t - base)
  uint32_t ra0, ra1, rb0, rb1;
  const uint32_t ua = lds0 + wm * O2_W4_UNIT, ub = lds0 + (2 + wn) * O2_W4_UNIT;
  const uint32_t kcoff0 = (uint32_t)(u * 128 + ((cg ^ sw) << 4)), kcoff1 = (uint32_t)(u * 128 + (((4 + cg) ^ sw) << 4));
  const uint32_t ksoff = (uint32_t)((4 * (cg & 1) + (u >> 2)) * 1056 + 2 * (cg >> 1) * 256 + 8 * (lane & 3));
  if (A_KC) { ra0 = ua + kcoff0; ra1 = ua + kcoff1; } else { ra0 = ua + ksoff; ra1 = ra0; }
  if (B_KC) { rb0 = ub + kcoff0; rb1 = ub + kcoff1; } else { rb0 = ub + ksoff; rb1 = rb0; }
  const uint32_t ta0 = 2 * ra0 + 4 * O2_W4_UNIT, ta1 = 2 * ra1 + 4 * O2_W4_UNIT, tb0 = 2 * rb0 + 4 * O2_W4_UNIT,
                 tb1 = 2 * rb1 + 4 * O2_W4_UNIT;
  // LDS-DMA sources: one per-lane byte offset per operand, a scalar offset per piece (p0 + t ps + h ph), K advance per K-tile
  uint32_t voa, vob, pa0, psa, pha, pb0, psb, phb, ka, kb;
  const int pr = lane >> 3, chunk = (lane & 7) ^ (pr & 7);                       // K-contiguous: piece row, swizzled 16-byte chunk
  const int rr = lane >> 4, cp = lane & 15;                                      // K-strided: k-row within the piece, 16-byte chunk
  const int sk0 = 32 * (wave >> 1) + 8 * (wave & 1);                             // K-strided: first k-row of this wave's pieces
  if (A_KC) { voa = (uint32_t)(pr * lda + chunk * 8) * 2u; pa0 = (uint32_t)(wave * 32 * lda) * 2u; psa = (uint32_t)(8 * lda) * 2u; pha = (uint32_t)(128 * lda) * 2u; ka = 128u; }
  else { voa = (uint32_t)((16 * (rr >> 1) + 4 * (rr & 1)) * lda + cp * 8) * 2u; pa0 = (uint32_t)(sk0 * lda) * 2u; psa = (uint32_t)lda * 2u; pha = 256u; ka = (uint32_t)(64 * lda) * 2u; }
  if (B_KC) { vob = (uint32_t)(pr * ldb + chunk * 8) * 2u; pb0 = (uint32_t)(wave * 32 * ldb) * 2u; psb = (uint32_t)(8 * ldb) * 2u; phb = (uint32_t)(128 * ldb) * 2u; kb = 128u; }
  else { vob = (uint32_t)((16 * (rr >> 1) + 4 * (rr & 1)) * ldb + cp * 8) * 2u; pb0 = (uint32_t)(sk0 * ldb) * 2u; psb = (uint32_t)ldb * 2u; phb = 256u; kb = (uint32_t)(64 * ldb) * 2u; }
  // descriptor bases (the statement builds the descriptors and advances their bases along K: every per-lane / per-piece
  // offset stays below 256 rows x row pitch)
  const bf16_t* abase = A_KC ? A + (size_t)m0 * lda : A + m0;
  const bf16_t* bbase = B_KC ? B + (size_t)n0 * ldb : B + n0;
  const uint32_t ldswa = lds0 + wave * 4 * (A_KC ? 1024 : 1056);
  const uint32_t ldswb = lds0 + 2 * O2_W4_UNIT + wave * 4 * (B_KC ? 1024 : 1056);
  const uint32_t nk = (uint32_t)(K / BK3);
  // cohort pacing inside the sweep (the TN statement only reads these): the counter's address for the ONE wave that arrives and
  // polls on the workgroup's behalf, 0 for the others and for unpaced launches
  const uint64_t pc64 = (wave == 0) ? (uint64_t)(uintptr_t)pace_ctr : 0ull;
  const uint64_t pcp = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pc64 >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pc64);          // (wave-uniform: SGPR operands)
  const uint32_t pcn = (uint32_t)__builtin_amdgcn_readfirstlane(pace_n);
#define O2_W4_OPERANDS                                                                                                   \
               [abase] "s"(abase), [bbase] "s"(bbase), [pa0] "s"(pa0), [psa] "s"(psa), [pha] "s"(pha), [pb0] "s"(pb0),       \
               [psb] "s"(psb), [phb] "s"(phb), [ldswa] "s"(ldswa), [ldswb] "s"(ldswb), [nk] "s"(nk), [ka] "s"(ka),       \
               [kb] "s"(kb), [ta0] "v"(ta0), [ta1] "v"(ta1), [tb0] "v"(tb0), [tb1] "v"(tb1), [pcp] "s"(pcp), [pcn] "s"(pcn)
  O2_W4_TRACE_POINT(0);
  if constexpr (!STAMP) {
#define O2_W4_RUN(STR)                                                                                                   \
  asm volatile(STR                                                                                                       \
               : [voa] "+v"(voa), [vob] "+v"(vob), [ra0] "+v"(ra0), [ra1] "+v"(ra1), [rb0] "+v"(rb0), [rb1] "+v"(rb1)   \
               : O2_W4_OPERANDS                                                                                          \
               : O2_W4_CLOBBERS)
    if constexpr (FORM == 0) O2_W4_RUN(O2_W4_ASM_NT);
    if constexpr (FORM == 1) O2_W4_RUN(O2_W4_ASM_NN);
    if constexpr (FORM == 2) O2_W4_RUN(O2_W4_ASM_TN);
    if constexpr (FORM == 3) O2_W4_RUN(O2_W4_ASM_TT);
#undef O2_W4_RUN
  }
  O2_W4_TRACE_POINT(1);
#ifdef O2_W4_STAMP
  if constexpr (STAMP) {
    unsigned t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    O2_TS(0);
#define O2_W4_RUNS(STR)                                                                                                  \
  asm volatile(STR                                                                                                       \
               : [voa] "+v"(voa), [vob] "+v"(vob), [ra0] "+v"(ra0), [ra1] "+v"(ra1), [rb0] "+v"(rb0), [rb1] "+v"(rb1),  \
                 [t0] "+s"(t0), [t1] "+s"(t1), [t2] "+s"(t2), [t3] "+s"(t3), [t4] "+s"(t4)                               \
               : O2_W4_OPERANDS                                                                                          \
               : O2_W4_CLOBBERS)
    if constexpr (FORM == 0) O2_W4_RUNS(O2_W4_ASM_NT_STAMP);
    if constexpr (FORM == 1) O2_W4_RUNS(O2_W4_ASM_NN_STAMP);
    if constexpr (FORM == 2) O2_W4_RUNS(O2_W4_ASM_TN_STAMP);
    if constexpr (FORM == 3) O2_W4_RUNS(O2_W4_ASM_TT_STAMP);
#undef O2_W4_RUNS
    O2_TS(1);
    if (blockIdx.x < 64 && lane == 0) {
      unsigned* d = o2_dbg_w4 + (blockIdx.x * 4 + wave) * 16;
      d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3; d[4] = t4; d[5] = nk;
    }
  }
#endif
#undef O2_W4_OPERANDS
  // epilogue: two passes through the LDS image (see w4_epilogue_rows / w4_generic_*)
  const uint32_t crow = lds0 + (uint32_t)(wm * 64 + u) * 1024u;
  const uint32_t be = crow + (uint32_t)(((wn * 32 + cg) ^ sw) << 4), bo = crow + (uint32_t)(((wn * 32 + 4 + cg) ^ sw) << 4);
  const bool lean = !epi.out_fp32 && !epi.save_pre && epi.act == 0 && !epi.residual && epi.thr == 0 && !epi.dgelu_pre &&
                    !epi.rowscale && epi.beta == 0.f && !epi.mul;
  const bool generic = !lean && !epi.out_fp32;
  const int ne = n0 + 8 * (tid & 31);
  float bias8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias8[k] = 0.f;
  if constexpr (EK != 0) {
    // compile-time epilogue kind (epi8_finish<EK>): the pass's 16 rows per lane as straight-line code, four rows interleaved;
    // what a row loads from memory (residual / factor) is ALL put in flight before the accumulators are staged -- 64 KiB per
    // CU against the 16 KiB of the runtime form, whose (load, LDS read, ~300 dependent vector instructions, store) chain per
    // two rows ran one wave per SIMD at a quarter of its vector rate (profiles/r04_gemm_epilogue_parts.txt)
    Epi ek = epi;
    if (EK == 2) ek.rs_tile = epi.rowscale ? epi.rowscale[m0 / epi.rows_per_scale] : 1.0f;
    if (EK == 1 || EK == 2) unpack8(*reinterpret_cast<const u32x4*>(epi.bias + ne), bias8);
    const int r8 = tid >> 5, q = tid & 31;
    constexpr int UNR = EK == 1 ? 4 : 16;              // (2 / 3: short rows, and ld[] must stay in registers; kind 1 is bound by
                                                       // vector issue: 1 / 2 / 4 / 8 / 16 rows interleaved are within +-1 %)
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // kind 3 with colsum_ws: this lane's 8 columns over its 32 rows
#pragma unroll 1
    for (int hh = 0; hh < 2; ++hh) {
      u32x4 ld[16];
      if (EK == 2 || EK == 3) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int lr = g * 8 + r8;
          const int m = m0 + (lr >> 6) * 128 + hh * 64 + (lr & 63);
          ld[g] = EK == 2 ? *reinterpret_cast<const u32x4*>(epi.residual + (size_t)m * epi.ldr + ne)
                          : *reinterpret_cast<const u32x4*>(epi.mul + (size_t)m * epi.ldc + ne);
        }
      }
      __syncthreads();
      if (hh == 0) asm volatile(O2_W4_CSTAGE0 : : [be] "v"(be), [bo] "v"(bo) : "memory");
      else asm volatile(O2_W4_CSTAGE1 : : [be] "v"(be), [bo] "v"(bo) : "memory");
      __syncthreads();
#pragma unroll UNR
      for (int g = 0; g < 16; ++g) {
        const int lr = g * 8 + r8;                       // (lr & 7) == r8
        const int m = m0 + (lr >> 6) * 128 + hh * 64 + (lr & 63);
        const char* row = smem + lr * 1024;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(row + (((2 * q) ^ r8) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(row + (((2 * q + 1) ^ r8) << 4));
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        Pre8 pq;
        if (EK == 2) pq.res = ld[g];
        if (EK == 3) pq.pre = ld[g];
        epi8_finish<EK>(ek, m, ne, v, bias8, pq);
        if (EK == 3 && epi.colsum_ws) {                  // (v holds the final fp32 values: sum what was STORED, i.e. their bf16 rounding)
          float r[8];
          unpack8(pack8(v), r);
#pragma unroll
          for (int k = 0; k < 8; ++k) csum[k] += r[k];
        }
      }
    }
    if (EK == 3 && epi.colsum_ws) {
      // the tile's column sums: 8 row groups (tid >> 5) hold partial sums of the same 256 columns -> through LDS (free now), added in
      // a fixed order (r8 = 0..7), one fp32 row of the workspace per tile row: no atomics, reproducible
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int k = 0; k < 8; ++k) red[(r8 * 32 + q) * 8 + k] = csum[k];
      __syncthreads();
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += red[(g * 32 + (tid >> 3)) * 8 + (tid & 7)];
      epi.colsum_ws[(size_t)(m0 >> 8) * epi.N + n0 + tid] = t;
    }
    return;
  }
  // (measured and not kept, round 6: the lean epilogue straight from the accumulator registers -- no LDS image, no barrier, 64
  // stores of 16 rows x 32 B per wave, bit-identical: 1-3 % SLOWER on the K = 3072 GEMMs, level on K = 12288; the 32-byte row
  // segments cost the store path more than the two LDS passes cost the workgroup.  profiles/r06_direct_epilogue_ab.txt)
  if (generic && epi.bias) unpack8(*reinterpret_cast<const u32x4*>(epi.bias + ne), bias8);
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    Pre8 qa[2], qb[2];
    if (generic) {
      w4_generic_load(epi, m0, ne, hh, tid, 0, qa);
      w4_generic_load(epi, m0, ne, hh, tid, 1, qb);
    }
    __syncthreads();                                 // pass 0: every wave's last LDS-DMA pieces have landed, nobody reads the stages
    if (hh == 0) asm volatile(O2_W4_CSTAGE0 : : [be] "v"(be), [bo] "v"(bo) : "memory");
    else asm volatile(O2_W4_CSTAGE1 : : [be] "v"(be), [bo] "v"(bo) : "memory");
    __syncthreads();
    if (hh == 0) { O2_TS(2); } else { O2_TS(4); }
    if (generic) {
#pragma unroll 1
      for (int it2 = 0; it2 < 4; ++it2) {
        w4_generic_finish(epi, smem, m0, ne, hh, tid, 2 * it2, bias8, qa);
        if (it2 < 3) w4_generic_load(epi, m0, ne, hh, tid, 2 * it2 + 2, qa);
        w4_generic_finish(epi, smem, m0, ne, hh, tid, 2 * it2 + 1, bias8, qb);
        if (it2 < 3) w4_generic_load(epi, m0, ne, hh, tid, 2 * it2 + 3, qb);
      }
    } else {
      w4_epilogue_rows(epi, smem, m0, n0, hh, tid);
    }
    if (hh == 0) { O2_TS(3); } else { O2_TS(5); }
  }
#ifdef O2_W4_STAMP
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    O2_TS(6);
    if (blockIdx.x < 64 && lane == 0) {
      unsigned* d = o2_dbg_w4 + (blockIdx.x * 4 + wave) * 16;
      for (int k = 0; k < 7; ++k) d[8 + k] = ts[k] - ts0;
    }
  }
#endif
#undef O2_TS
}

template <int FORM, bool STAMP, int EK = 0>
__global__ __launch_bounds__(256, 1) void gemm256w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                          int M, int N, int K, int lda, int ldb, int tiles_m,
                                                          int tiles_n, Epi epi) {
  __shared__ __attribute__((aligned(16))) char smem[8 * O2_W4_UNIT];
  gemm256w_tile<FORM, STAMP, EK>(A, B, M, N, K, lda, ldb, tiles_m, tiles_n, epi, w4_tile_id(), smem);
}

// the epilogue kind of epi8_finish a whole-tile bf16 problem qualifies for (0: the runtime form)
static int w4_epi_kind(const Epi& e) {
  if (e.out_fp32 || e.save_pre || e.dgelu_pre || e.beta != 0.f || e.colscale_n > 0 || e.res_first || e.res_mod > 0 || e.act == 2) return 0;
  // (thr == 0 keeps everything at scale 256 / 256 and an absent row scale is a multiplication by 1.0f: both exact, so kinds 1 / 2
  // also serve the layers without dropout or without a DropPath scale)
  if (e.bias && e.act == 1 && e.save_dact && !e.residual && !e.rowscale && !e.mul) return 1;
  if (e.bias && e.act == 0 && e.residual && (!e.rowscale || e.rows_per_scale % 256 == 0) && !e.mul && !e.save_dact &&
      e.ldr % 8 == 0 && !((uintptr_t)e.residual & 15))
    return 2;
  if (!e.bias && e.act == 0 && !e.thr && !e.residual && !e.rowscale && e.mul && !e.save_dact) return 3;
  return 0;
}

template <int FORM>
__global__ __launch_bounds__(256, 1) void gemm256w_grouped_kernel(GArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[8 * O2_W4_UNIT];
  const int id = w4_tile_id();
  int pace_n = 0;
  const unsigned int* pace_ctr = g.pace ? w4_cohort_start(pace_n) : nullptr;
  if (g.pace < 2) pace_ctr = nullptr;                                   // pace 1: the start barrier only
  int pi = 0;
  while (pi + 1 < g.n && id >= g.p[pi].tile_end) ++pi;
  const int first = pi ? g.p[pi - 1].tile_end : 0;
  const GProb& P = g.p[pi];
  const Epi epi = P.epi;
  gemm256w_tile<FORM, false>(P.A, P.B, P.M, P.N, P.K, P.lda, P.ldb, P.tiles_m, P.tiles_n, epi, id - first, smem, pace_ctr, pace_n);
}

// ------------------------------------------------------------------------------------------
// small fp32 GEMM (table algebra; sizes ~ [115 x D] x [D x D]): 64x64 tile, 16x16 threads, 4x4 micro-tile
// ------------------------------------------------------------------------------------------
// BK: k-depth of one staged step.  The skinny form is a serial chain of (global load -> LDS -> barrier) steps whose
// cost is the load latency, not the 16 FMAs: BK = 64 makes the chain 4x shorter with 4x the bytes in flight per step.
template <int RM, int BK = 16>   // block tile = (16*RM) x 64; RM = 4 (64 rows) or 1 (16 rows: skinny-M problems fill the chip)
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                    float* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                    int ldc, int ta, int tb, float alpha, float beta, int kchunk,
                                                    size_t zstride) {
  // split-K: block z multiplies k in [z*kchunk, (z+1)*kchunk) into its own [M, N] slab at C + z*zstride (alpha = 1,
  // beta = 0 there; sgemm_reduce_kernel sums the slabs).  One slab (kchunk = K) is the plain product.
  const int kbase = blockIdx.z * kchunk;
  const int Kend = (kbase + kchunk) < K ? (kbase + kchunk) : K;
  C += (size_t)blockIdx.z * zstride;
  constexpr int TMB = 16 * RM;
  __shared__ float sA[BK][TMB + 1];
  __shared__ float sB[BK][64 + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * TMB, n0 = blockIdx.x * 64;
  float acc[RM][4] = {};
  // register double-buffering: the global loads of step k+1 are in flight while step k is multiplied out of LDS (and
  // all loads of a step are issued before the first LDS store: a fused load/store loop would serialise one memory
  // round trip per element)
  constexpr int NA = BK * TMB / 256, NB = BK * 64 / 256;
  float ra[NA], rb[NB];
  auto load_step = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = threadIdx.x + i * 256;
      int kk, mm;
      if (ta) { mm = e % TMB; kk = e / TMB; } else { kk = e % BK; mm = e / BK; }
      const int gm = m0 + mm, gk = k0 + kk;
      ra[i] = (gm < M && gk < Kend) ? (ta ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = threadIdx.x + i * 256;
      int kk, nn;
      if (tb) { kk = e % BK; nn = e / BK; } else { nn = e & 63; kk = e >> 6; }
      const int gn = n0 + nn, gk = k0 + kk;
      rb[i] = (gn < N && gk < Kend) ? (tb ? B[(size_t)gn * ldb + gk] : B[(size_t)gk * ldb + gn]) : 0.f;
    }
  };
  load_step(kbase);
  for (int k0 = kbase; k0 < Kend; k0 += BK) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = threadIdx.x + i * 256;
      if (ta) sA[e / TMB][e % TMB] = ra[i]; else sA[e % BK][e / BK] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = threadIdx.x + i * 256;
      if (tb) sB[e % BK][e / BK] = rb[i]; else sB[e >> 6][e & 63] = rb[i];
    }
    __syncthreads();
    if (k0 + BK < Kend) load_step(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      float a[RM], b[4];
#pragma unroll
      for (int i = 0; i < RM; ++i) a[i] = sA[kk][ty * RM + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = sB[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const int m = m0 + ty * RM + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float* c = C + (size_t)m * ldc + n;
      *c = alpha * acc[i][j] + (beta != 0.f ? beta * *c : 0.f);
    }
  }
}

// C = alpha * sum_z ws[z] + beta * C   (deterministic combine of the split-K slabs)
__global__ __launch_bounds__(256) void sgemm_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int M,
                                                           int N, int ldc, int S, float alpha, float beta) {
  const int64_t n = (int64_t)M * N;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += ws[(size_t)z * n + e];
    float* c = C + (size_t)(e / N) * ldc + (e % N);
    *c = alpha * s + (beta != 0.f ? beta * *c : 0.f);
  }
}

}  // namespace

O2_DEFINE_SALT_OP(gemm)

extern "C" int orbit2_abi_version(void) { return ORBIT2_ABI_VERSION; }

#ifdef O2_STAMP
extern "C" int orbit2_debug_read(unsigned int* host_dst, int n) {   // diagnostic build only
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(o2_dbg), sizeof(unsigned) * (size_t)n);
}
#endif

static int gemm_make_epi(const orbit2_gemm_args* a, Epi& e) {
  if (!a || !a->A || !a->B || !a->C) return O2_ERR_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return O2_ERR_ARG;
  // rows of a K-contiguous operand are clamped at staging and masked at the store, so M is free there; an operand
  // stored K-strided is fetched in 8-element pieces along its M / N axis; stores are 8 (bf16) / 4 (fp32) n wide
  // K: a K-contiguous operand is fetched in 8-element pieces along k (K % 8); K-strided operands take any K (whole
  // rows).  The 128-tile kernel stages a ragged K tail from a zero page; the ring kernel needs K % 32 == 0.
  if (((a->a_kc || a->b_kc) && a->K % 8) || a->N % 8 || (!a->a_kc && a->M % 8) || a->lda % 8 || a->ldb % 8 ||
      a->ldc % 4)
    return O2_ERR_ARG;
  if (a->tile_hint >= 256 && a->tile_hint <= 258 && a->K % BK3) return O2_ERR_ARG;   // the 8-phase kernel: whole 64-deep K-tiles
  if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15) return O2_ERR_ARG;
  if (a->drop_p < 0.f || a->drop_p >= 1.f) return O2_ERR_ARG;
  if (a->rowscale && a->rows_per_scale <= 0) return O2_ERR_ARG;
  if (a->colscale_n < 0 || a->colscale_n % 8) return O2_ERR_ARG;
  if (a->save_dact && (a->act != 1 || a->out_fp32)) return O2_ERR_ARG;   // the factor of a GELU output element; bf16 path
  // save_dact stores GELU'(pre) x dropout scale as q14 fixed point, range [-2, 2): max GELU' = 1.129, so the scale must stay
  // below 1.77 (drop_p < 0.434); above that the caller uses save_pre / dgelu_pre (climate_learn/_ops.py does)
  if (a->save_dact && 1.13f * (256.0f / (256.0f - (float)(unsigned)(a->drop_p * 256.0f + 0.5f))) >= 2.0f) return O2_ERR_UNSUPPORTED;
  if (a->mul && a->dgelu_pre) return O2_ERR_ARG;                           // one or the other (they share a load slot)
  if (((uintptr_t)a->save_dact | (uintptr_t)a->mul) & 15) return O2_ERR_ARG;
  e.colscale_n = a->colscale_n;
  e.colscale = a->colscale;
  e.save_dact = (bf16_t*)a->save_dact;
  e.mul = (const bf16_t*)a->mul;
  e.bias = (const bf16_t*)a->bias;
  e.save_pre = (bf16_t*)a->save_pre;
  e.dgelu_pre = (const bf16_t*)a->dgelu_pre;
  e.rowscale = a->rowscale;
  e.residual = (const bf16_t*)a->residual;
  e.C = a->C;
  e.seed = a->seed;
  e.M = a->M; e.N = a->N; e.ldc = a->ldc; e.ldr = a->ldr; e.res_mod = a->res_mod; e.res_first = a->res_first;
  e.rows_per_scale = a->rows_per_scale; e.act = a->act; e.out_fp32 = a->out_fp32;
  e.thr = (unsigned)(a->drop_p * 256.0f + 0.5f);
  e.dscale = 256.0f / (256.0f - (float)e.thr);
  e.beta = a->beta;
  e.rs_tile = 1.0f;
  e.colsum_ws = a->colsum_ws;
  return O2_OK;
}

extern "C" int orbit2_gemm_bf16_grouped(const orbit2_gemm_args* args, int n, void* stream) {
  if (!args || n <= 0 || n > ORBIT2_GEMM_MAX_GROUP) return O2_ERR_ARG;
  if (n == 1) return orbit2_gemm_bf16(args, stream);
  for (int i = 0; i < n; ++i)
    if (args[i].colsum_ws) return O2_ERR_UNSUPPORTED;      // single launches only (orbit2_gemm_bf16_colsum_rows)
  GArgs g;
  g.n = n;
  g.pace = 0;
  // 256-tile 8-phase kernel when every problem of the group can take it and the group fills the chip; 128-tile otherwise
  bool big = args[0].tile_hint != 128;
  long t256 = 0;
  for (int i = 0; i < n; ++i) {
    const orbit2_gemm_args* a = args + i;
    if (a->K % BK3 || a->K < 2 * BK3 || a->M < 256 || a->N < 256) big = false;
    t256 += (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
  }
  if (args[0].tile_hint < 256 && t256 < 192) big = false;
  const int TB = big ? 256 : 128;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const orbit2_gemm_args* a = args + i;
    if (a->a_kc != args[0].a_kc || a->b_kc != args[0].b_kc) return O2_ERR_ARG;   // one operand form per group
    const int rc = gemm_make_epi(a, g.p[i].epi);
    if (rc) return rc;
    GProb& P = g.p[i];
    P.A = (const bf16_t*)a->A; P.B = (const bf16_t*)a->B;
    P.M = a->M; P.N = a->N; P.K = a->K; P.lda = a->lda; P.ldb = a->ldb;
    P.tiles_m = (a->M + TB - 1) / TB; P.tiles_n = (a->N + TB - 1) / TB;
    total += P.tiles_m * P.tiles_n;
    P.tile_end = total;
  }
  hipStream_t s = (hipStream_t)stream;
  if (big) {
    bool whole = args[0].tile_hint != 256;           // the 4-wave kernel takes whole tiles only (hint 256 forces the 8-phase kernel)
    for (int i = 0; i < n; ++i) whole = whole && args[i].M % 256 == 0 && args[i].N % 256 == 0;
    if (whole) {
      dim3 grid(total), block(256);
      // cohort start barrier: every problem's tiles sweep >= 512 K-tiles (a round lasts >= 0.6 ms: the bounded wait is noise
      // against it, and a lost cohort costs a whole sweep of re-fetched strips)
      // ORBIT2_W4_PACE: 0 off, 1 (default) the cohort start barrier, 2 + the check points inside the sweep.  2 brings the launch's
      // bytes beyond L2 to the tile-walk bound (48 GB, 3.7 x algorithmic; 1.71 GHz stand-alone) and -2.3 % stand-alone, but the
      // SAME time in the step (166.6 vs 166.6 ms per step, profiles/r06_pace2b_instep.txt): every column group then sweeps at the
      // slowest group's rate, which is what a round takes anyway.  Kept selectable, not default.
      static const int pace_env = [] { const char* e = getenv("ORBIT2_W4_PACE"); return e ? atoi(e) : 1; }();
      int kmin = args[0].K;
      for (int i = 1; i < n; ++i) kmin = args[i].K < kmin ? args[i].K : kmin;
      g.pace = (O2_W4_WALK && pace_env && kmin >= 512 * BK3 && total > 256) ? pace_env : 0;
      if (args[0].a_kc && args[0].b_kc) hipLaunchKernelGGL((gemm256w_grouped_kernel<0>), grid, block, 0, s, g);
      else if (args[0].a_kc) hipLaunchKernelGGL((gemm256w_grouped_kernel<1>), grid, block, 0, s, g);
      else if (args[0].b_kc) hipLaunchKernelGGL((gemm256w_grouped_kernel<3>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm256w_grouped_kernel<2>), grid, block, 0, s, g);
      O2_CHECK_LAUNCH();
      return O2_OK;
    }
    dim3 grid(total), block(512);
    if (args[0].a_kc && args[0].b_kc) hipLaunchKernelGGL((gemm256t_grouped_kernel<true, true>), grid, block, 0, s, g);
    else if (args[0].a_kc) hipLaunchKernelGGL((gemm256t_grouped_kernel<true, false>), grid, block, 0, s, g);
    else if (args[0].b_kc) hipLaunchKernelGGL((gemm256t_grouped_kernel<false, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm256t_grouped_kernel<false, false>), grid, block, 0, s, g);
    O2_CHECK_LAUNCH();
    return O2_OK;
  }
  dim3 grid(total), block(256);
  if (args[0].a_kc && args[0].b_kc) hipLaunchKernelGGL((gemm128_grouped_kernel<true, true>), grid, block, 0, s, g);
  else if (args[0].a_kc) hipLaunchKernelGGL((gemm128_grouped_kernel<true, false>), grid, block, 0, s, g);
  else if (args[0].b_kc) hipLaunchKernelGGL((gemm128_grouped_kernel<false, true>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm128_grouped_kernel<false, false>), grid, block, 0, s, g);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

// does this call run on the 4-wave kernel with epilogue kind 3 (the only path that fills colsum_ws)?  Mirrors the dispatch below.
static bool gemm_fuses_colsum(const orbit2_gemm_args* a, const Epi& e) {
  if (!(a->a_kc && !a->b_kc) || a->M % 256 || a->N % 256 || a->K % 64 || a->K < 128 || w4_epi_kind(e) != 3) return false;
  if (a->tile_hint == 260) return true;
  if (a->tile_hint != 0) return false;
  const long t256 = (long)(a->M / 256) * (a->N / 256);
  const long rounds = (t256 + 255) / 256;
  return t256 >= 192 && (double)t256 / (double)(rounds * 256) >= 0.70;
}

extern "C" int orbit2_gemm_bf16_colsum_rows(const orbit2_gemm_args* a) {
  Epi e;
  if (!a || gemm_make_epi(a, e)) return 0;
  return gemm_fuses_colsum(a, e) ? a->M / 256 : 0;
}

extern "C" int orbit2_gemm_bf16(const orbit2_gemm_args* a, void* stream) {
  Epi e;
  const int rc_epi = gemm_make_epi(a, e);
  if (rc_epi) return rc_epi;
  if (a->colsum_ws && !gemm_fuses_colsum(a, e)) return O2_ERR_UNSUPPORTED;   // ask orbit2_gemm_bf16_colsum_rows first
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* A = (const bf16_t*)a->A;
  const bf16_t* B = (const bf16_t*)a->B;
  // tile choice: the 256^2 8-phase kernel (1 workgroup/CU) needs enough tiles to fill the chip; the 128^2 kernel
  // (2 workgroups/CU) takes small or ragged problems.  tile_hint forces one (tests / tuning).
  const long t256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
  int tile = a->tile_hint;
  if (tile == 257 || tile == 258) tile = 256;       // hints of the round-2 A/B tools: the same kernel
  if (tile == 64) tile = 128;                         // the 128-wide kernel on 64-row tiles (see the dispatch at the end)
  if (tile != 128 && tile != 256 && tile != 260 && tile != 261 && tile != 262) {
    // measured on MI355X (tools/gemm_p8_ab.py, tools/gemm_t8_ab.py, tools/gemm_w4_ab.py; profiles/r02_gemm_*, r03_gemm_w4_*): a
    // 256-tile kernel wins in every operand form whenever its tiles fill the chip -- the 4-wave kernel on whole tiles
    // (+7 ... +16 % over the 8-phase kernel), the 8-phase kernel on ragged M / N; the 128^2 kernel (2 workgroups/CU, ragged K)
    // takes small or ragged problems
    const long rounds = (t256 + 255) / 256;
    const double util = (double)t256 / (double)(rounds * 256);
    tile = (a->K % BK3 == 0 && a->K >= 2 * BK3 && a->M >= 256 && a->N >= 256 && t256 >= 192 && util >= 0.70) ? 256 : 128;
    // the 4-wave kernel: whole tiles, and an epilogue that is not vector-ALU heavy -- one wave per SIMD issues vector
    // instructions at half the rate two waves reach, so GELU / GELU' / dropout epilogues (~35 instructions per value) cost it
    // more than its main loop gains (profiles/r03_gemm_w4_epilogues.txt); K-contiguous operands whose rows are BOTH a
    // multiple of 8 KiB apart (fc2-shaped: every row's k-offset on the same memory channel) also stay on the 8-phase kernel
    // (the three hot heavy combinations have compile-time epilogues there -- w4_epi_kind -- and do go to it)
    const int form_ = a->a_kc ? (a->b_kc ? 0 : 1) : (a->b_kc ? 3 : 2);
    const int ek_ = w4_epi_kind(e);
    const bool heavy = (e.act != 0 || e.thr != 0 || e.dgelu_pre != nullptr) && !((ek_ == 1 || ek_ == 2) && form_ == 0);
    const bool camped = a->a_kc && a->b_kc && a->K >= 8192 && a->lda % 4096 == 0 && a->ldb % 4096 == 0;
    if (tile == 256 && a->M % 256 == 0 && a->N % 256 == 0 && !heavy && !camped) tile = 260;
  }
  const bool w4_runtime_epi = tile == 262;             // hint 262: the 4-wave kernel with the runtime epilogue (A/B, bit-identity tests)
  if (tile == 262) tile = 260;
  if (tile == 260 || tile == 261) {                   // 4-wave kernel: whole tiles (261: its stamped diagnostic form)
    if (a->M % 256 || a->N % 256 || a->K % BK3) return O2_ERR_ARG;
    const int tiles_m = a->M / BM2, tiles_n = a->N / BN2;
    const dim3 grid(tiles_m * tiles_n), block(256);
    const int form = a->a_kc ? (a->b_kc ? 0 : 1) : (a->b_kc ? 3 : 2);
#define O2_W4_LAUNCH(F, S) hipLaunchKernelGGL((gemm256w_kernel<F, S>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb, tiles_m, tiles_n, e)
    const int ek = w4_runtime_epi ? 0 : w4_epi_kind(e);
#define O2_W4_LAUNCH_EK(F, EKV) hipLaunchKernelGGL((gemm256w_kernel<F, false, EKV>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb, tiles_m, tiles_n, e)
    if (tile == 260 && form == 0 && ek == 1) { O2_W4_LAUNCH_EK(0, 1); }
    else if (tile == 260 && form == 0 && ek == 2) { O2_W4_LAUNCH_EK(0, 2); }
    else if (tile == 260 && form == 1 && ek == 3) { O2_W4_LAUNCH_EK(1, 3); }
    else if (tile == 260) {
      switch (form) {
        case 0: O2_W4_LAUNCH(0, false); break;
        case 1: O2_W4_LAUNCH(1, false); break;
        case 2: O2_W4_LAUNCH(2, false); break;
        default: O2_W4_LAUNCH(3, false); break;
      }
    } else {
#ifdef O2_W4_STAMP
      switch (form) {
        case 0: O2_W4_LAUNCH(0, true); break;
        case 1: O2_W4_LAUNCH(1, true); break;
        case 2: O2_W4_LAUNCH(2, true); break;
        default: O2_W4_LAUNCH(3, true); break;
      }
#else
      return O2_ERR_UNSUPPORTED;
#endif
    }
#undef O2_W4_LAUNCH
#undef O2_W4_LAUNCH_EK
    O2_CHECK_LAUNCH();
    return O2_OK;
  }
  if (tile == 256) {
    const int tiles_m = (a->M + BM2 - 1) / BM2, tiles_n = (a->N + BN2 - 1) / BN2;
    dim3 grid(tiles_m * tiles_n), block(512);
    if (a->a_kc && a->b_kc)
      hipLaunchKernelGGL((gemm256t_kernel<true, true>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb,
                         tiles_m, tiles_n, e);
    else if (a->a_kc && !a->b_kc)
      hipLaunchKernelGGL((gemm256t_kernel<true, false>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb,
                         tiles_m, tiles_n, e);
    else if (!a->a_kc && a->b_kc)
      hipLaunchKernelGGL((gemm256t_kernel<false, true>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb,
                         tiles_m, tiles_n, e);
    else
      hipLaunchKernelGGL((gemm256t_kernel<false, false>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb,
                         tiles_m, tiles_n, e);
    O2_CHECK_LAUNCH();
    return O2_OK;
  }
  int tiles_m = (a->M + BM - 1) / BM;
  const int tiles_n = (a->N + BN - 1) / BN;
  // fewer 128 x 128 tiles than two per CU and a K-contiguous A: 64-row tiles, so that every CU still holds two workgroups
  // (hint 64 forces them, hint 128 forbids them)
  if (a->tile_hint == 64 && !a->a_kc) return O2_ERR_ARG;
  const bool half_rows = a->a_kc && (a->tile_hint == 64 || (a->tile_hint != 128 && (long)tiles_m * tiles_n < 2 * 256 && a->M > 64));
  if (half_rows) {
    tiles_m = (a->M + 63) / 64;
    dim3 grid(tiles_m * tiles_n), block(256);
    if (a->b_kc)
      hipLaunchKernelGGL((gemm128_kernel<true, true, 64>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb, tiles_m,
                         tiles_n, e);
    else
      hipLaunchKernelGGL((gemm128_kernel<true, false, 64>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb, tiles_m,
                         tiles_n, e);
    O2_CHECK_LAUNCH();
    return O2_OK;
  }
  dim3 grid(tiles_m * tiles_n), block(256);
  if (a->a_kc && a->b_kc)
    hipLaunchKernelGGL((gemm128_kernel<true, true>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb,
                       tiles_m, tiles_n, e);
  else if (a->a_kc && !a->b_kc)
    hipLaunchKernelGGL((gemm128_kernel<true, false>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb,
                       tiles_m, tiles_n, e);
  else if (!a->a_kc && a->b_kc)
    hipLaunchKernelGGL((gemm128_kernel<false, true>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb,
                       tiles_m, tiles_n, e);
  else
    hipLaunchKernelGGL((gemm128_kernel<false, false>), grid, block, 0, s, A, B, a->M, a->N, a->K, a->lda, a->ldb,
                       tiles_m, tiles_n, e);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

// split-K plan of the skinny table products: enough slabs to put >= ~1024 workgroups on the chip, >= 4 staged k-steps each
static int sgemm_splits(int M, int N, int K) {
  const long blocks = (long)((N + 63) / 64) * (M > 32 ? (M + 63) / 64 : (M + 15) / 16);
  if (blocks >= 512 || K < 1024) return 1;
  int s = (int)((1024 + blocks - 1) / blocks);
  const int maxs = K / 256;
  s = s < maxs ? s : maxs;
  return s < 1 ? 1 : (s > 16 ? 16 : s);
}

extern "C" int64_t orbit2_sgemm_f32_ws_floats(int M, int N, int K) {
  const int s = sgemm_splits(M, N, K);
  return s > 1 ? (int64_t)s * M * N : 0;
}

extern "C" int orbit2_sgemm_f32_ws(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                                   int ldc, int ta, int tb, float alpha, float beta, float* ws, int64_t ws_floats,
                                   void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return O2_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long blocks64 = (long)((N + 63) / 64) * ((M + 63) / 64);
  int S = sgemm_splits(M, N, K);
  if (S > 1 && (!ws || ws_floats < (int64_t)S * M * N)) S = 1;      // no workspace: plain product
  int kchunk = K;
  if (S > 1) kchunk = (((K + S - 1) / S) + 63) / 64 * 64;
  float* out = S > 1 ? ws : C;
  const int ldo = S > 1 ? N : ldc;
  const float al = S > 1 ? 1.f : alpha, be = S > 1 ? 0.f : beta;
  const size_t zs = (size_t)M * N;
  if (blocks64 >= 512) {
    dim3 grid((N + 63) / 64, (M + 63) / 64, 1), block(256);
    hipLaunchKernelGGL((sgemm_kernel<4, 16>), grid, block, 0, st, A, B, C, M, N, K, lda, ldb, ldc, ta, tb, alpha, beta, K,
                       (size_t)0);
  } else if (M > 32) {
    dim3 grid((N + 63) / 64, (M + 63) / 64, S), block(256);
    hipLaunchKernelGGL((sgemm_kernel<4, 32>), grid, block, 0, st, A, B, out, M, N, K, lda, ldb, ldo, ta, tb, al, be, kchunk, zs);
  } else {
    dim3 grid((N + 63) / 64, (M + 15) / 16, S), block(256);
    hipLaunchKernelGGL((sgemm_kernel<1, 64>), grid, block, 0, st, A, B, out, M, N, K, lda, ldb, ldo, ta, tb, al, be, kchunk, zs);
  }
  O2_CHECK_LAUNCH();
  if (S > 1 && blocks64 < 512) {
    hipLaunchKernelGGL(sgemm_reduce_kernel, dim3((unsigned)(((int64_t)M * N + 255) / 256 < 4096 ? ((int64_t)M * N + 255) / 256 : 4096)), dim3(256), 0, st, ws, C, M, N, ldc, S, alpha,
                       beta);
    O2_CHECK_LAUNCH();
  }
  return O2_OK;
}

extern "C" int orbit2_sgemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                int ta, int tb, float alpha, float beta, void* stream) {
  return orbit2_sgemm_f32_ws(A, B, C, M, N, K, lda, ldb, ldc, ta, tb, alpha, beta, nullptr, 0, stream);
}
