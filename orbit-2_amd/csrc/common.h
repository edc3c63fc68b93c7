// Shared device helpers for liborbit2_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define O2_OK 0
#define O2_ERR_ARG (-1)
#define O2_ERR_LAUNCH (-2)
#define O2_ERR_UNSUPPORTED (-3)

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
// two floats -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32.  (The scalar form -- two conversions, a mask
// and a shift-or -- cost four instructions per pair in every epilogue and elementwise kernel that stores bf16; same rounding,
// same bits.)
typedef __attribute__((ext_vector_type(2))) float o2_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 o2_bf16x2;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const o2_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, o2_bf16x2));
}

// 16-byte LDS-DMA: every lane supplies its own global source; the LDS destination is
// (wave-uniform base) + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(void, gsrc), LDS_PTR(void, lds_wave_base), 16, 0, 0);
}
// 4-byte LDS-DMA: lane l's dword lands at (wave-uniform base) + 4 l  (per-row statistics tables of the attention backward)
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(void, gsrc), LDS_PTR(void, lds_wave_base), 4, 0, 0);
}
// The same LDS-DMA as an inline-asm statement the compiler cannot see into (saddr form: wave-uniform 64-bit base + 32-bit
// per-lane byte offset).  hipcc models the builtin above as an LDS write pending on vmcnt and, when it cannot prove that an LDS
// read touches other bytes (the other stage of the same __shared__ array, addressed through a lane-dependent register), puts
// s_waitcnt vmcnt(0) in front of the read: a tile prefetched at the top of a loop body is drained in the MIDDLE of that body
// (in front of the first ds_read_b64_tr_b16) instead of at its end.  With this form the kernel owns the ordering -- its own
// s_waitcnt vmcnt(0) + s_barrier before the first read of the slot -- and the compiler's vmcnt arithmetic stays safe: it can only
// under-count the operations younger than one of ITS loads, i.e. wait longer than needed.  M0 (the DMA's LDS base) is written
// and restored inside the statement.
__device__ __forceinline__ void glds16_asm(const void* gbase_uniform, uint32_t lane_byte_off, uint32_t lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_byte_off), "s"(gbase_uniform), "s"(lds_dst_uniform)
               : "memory");
}
// 4-byte form of the same (per-row statistics tables): lane l's dword from gbase + 4 l lands at lds_dst + 4 l
__device__ __forceinline__ void glds4_asm(const void* gbase_uniform, uint32_t lane_byte_off, uint32_t lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_byte_off), "s"(gbase_uniform), "s"(lds_dst_uniform)
               : "memory");
}
// transposed 4x16 block read (ds_read_b64_tr_b16): lane i of a 16-lane group receives column i.
__device__ __forceinline__ bf16x4 lds_tr4(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4, lds_addr));
}

// ---- device-side seed salt --------------------------------------------------------------
// Every seeded kernel xors this word into the seed it was launched with.  It is 0 unless a caller bumps it
// (orbit2_seed_salt): a training step captured in a hipGraph has its kernel arguments -- the seeds -- frozen, so the graph
// starts with a one-thread kernel that advances the salt and every replay draws fresh dropout / DropPath masks while
// forward and backward of the same replay still agree.  One copy per translation unit (no relocatable device code);
// orbit2_seed_salt updates all of them.
static __device__ uint64_t o2_seed_salt = 0;
#define O2_DEFINE_SALT_OP(tag)                                                                           \
  namespace {                                                                                            \
  __global__ void o2_salt_kernel_##tag(uint64_t v, int add) { o2_seed_salt = add ? o2_seed_salt + v : v; } \
  }                                                                                                      \
  void o2_salt_op_##tag(uint64_t v, int add, hipStream_t s) {                                            \
    hipLaunchKernelGGL(o2_salt_kernel_##tag, dim3(1), dim3(1), 0, s, v, add);                            \
  }
void o2_salt_op_gemm(uint64_t v, int add, hipStream_t s);
void o2_salt_op_attn(uint64_t v, int add, hipStream_t s);
void o2_salt_op_elem(uint64_t v, int add, hipStream_t s);

// ---- counter-based dropout hash ------------------------------------------------------
// One 32-bit hash per group of 4 consecutive elements; element e of the group uses byte e.
// keep <=> byte >= thr   (thr = round(p*256), effective drop prob thr/256, scale 256/(256-thr)).
// tests/ replicate this function in numpy to build bit-identical masks for the oracle.
__device__ __host__ __forceinline__ uint32_t o2_hash(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx_lo,
                                                      uint32_t idx_hi) {
  // fold the (rarely non-zero) high words with shifts/adds only, then the two-multiply "lowbias32" finaliser:
  // v_mul_lo_u32 is quarter-rate on CDNA4, and this hash sits in the attention inner loops
  const uint32_t t = idx_hi ^ seed_hi;
  uint32_t h = idx_lo ^ seed_lo ^ ((t << 16) | (t >> 16)) ^ (t + (t << 3));
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t o2_hash64(uint64_t seed, uint64_t idx) {
  return o2_hash((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)idx, (uint32_t)(idx >> 32));
}

// ---- attention-probability dropout: a factored hash ------------------------------------------------------------
// keep(row, key) is byte (key & 3) of  mix(R(row) ^ K(key >> 2))  >= thr, with R and K full o2_hash64 values of the
// global query row (b*H + head)*L + q and of the key group.  In every kernel one of the two factors is fixed per lane
// for the whole launch (the query row in the forward / dQ kernels, the key group in the dK / dV kernels) and the other
// is shared by the wave, so the per-element-group cost is one xor, one multiply and one xor-shift instead of a 64-bit
// index and two multiplies.  The single multiply after the xor removes the GF(2) structure of R ^ K (without it the
// four masks of any 2x2 row/key-group rectangle would be linearly dependent); tests/hashmask.py holds the numpy
// replica and the statistics the choice was checked with.
#define O2_ATTN_KEY_SALT 0x85EBCA6B9E3779B9ull
__device__ __forceinline__ uint32_t o2_attn_rowhash(uint64_t seed, uint64_t row) { return o2_hash64(seed, row); }
__device__ __forceinline__ uint32_t o2_attn_keyhash(uint64_t seed, uint32_t kg) {
  return o2_hash64(seed ^ O2_ATTN_KEY_SALT, (uint64_t)kg);
}
__device__ __forceinline__ uint32_t o2_attn_mix(uint32_t r, uint32_t k) {
  uint32_t x = (r ^ k) * 0x9E3779B1u;
  return x ^ (x >> 16);
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// GELU / GELU' for the bf16 GEMM epilogues: Phi(x) through erfc's rational-exponential form (Abramowitz & Stegun
// 7.1.26, |abs err| <= 1.5e-7, evaluated on |x| so the negative tail has no cancellation): one v_rcp + one v_exp
// + 7 FMAs instead of ocml erff (~40 instructions).  The results are rounded to bf16 (quantum >= 2^-9 relative),
// the fp32 image path (conv GELU) keeps the erff forms above.
__device__ __forceinline__ float o2_half_erfc_abs(float x, float& e) {   // 0.5*erfc(|x|/sqrt2), e = exp(-x^2/2)
  const float u = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);
  return 0.5f * p * t * e;
}
__device__ __forceinline__ float gelu_fast(float x) {
  float e;
  const float y = o2_half_erfc_abs(x, e);
  return x * (x > 0.f ? 1.0f - y : y);
}
// GELU(x) and GELU'(x) from one evaluation of the shared terms (same expressions as gelu_fast / dgelu_fast: same bits)
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
  float e;
  const float y = o2_half_erfc_abs(x, e);
  const float cdf = x > 0.f ? 1.0f - y : y;
  g = x * cdf;
  dg = cdf + x * 0.3989422804014327f * e;
}
__device__ __forceinline__ float dgelu_fast(float x) {
  float e;
  const float y = o2_half_erfc_abs(x, e);
  return (x > 0.f ? 1.0f - y : y) + x * 0.3989422804014327f * e;
}

// the GELU-backward factor tensor (save_dact / mul): signed 16-bit fixed point, 14 fraction bits.  The factor lies in
// [-0.15, 1.26] (GELU' in [-0.13, 1.13] times the dropout scale): q14 keeps it to 3e-5 where bf16 would keep 4e-3 near 1 --
// the error of this factor goes un-averaged into the residual-gradient stream (tests/test_model_gpu.py: pos_embed).
// pack_q14 SATURATES at the ends of [-2, 2): a factor outside it (GELU' up to 1.13 x a dropout scale >= 1.77, i.e. drop_p >= 0.434 --
// rejected by gemm_make_epi) must never wrap to the opposite sign.
// Round-to-nearest-even and the saturation without an integer conversion: x * 2^14 + 1.5 * 2^23 is exact up to the one rounding at
// ulp = 1 (the scaling is a power of two), which leaves the integer, two's complement, in the low mantissa bits; clamping the SUM to
// [magic - 32768, magic + 32767] saturates it (the bounds are integers: clamp and rounding commute), and one byte permute packs the
// two low halves.  Same bits as  min(max(rint(x * 16384), -32768), 32767)  for every finite x (NaN packs as 0x8000... never produced:
// the factor is bounded); 2.5 vector instructions per value instead of 5.5 in the GELU epilogue that is bound by them.
__device__ __forceinline__ unsigned pack_q14(float lo, float hi) {
  constexpr float MAGIC = 12582912.0f;                 // 1.5 * 2^23 = 0x4B400000: low 16 bits zero
  const float a = __builtin_amdgcn_fmed3f(fmaf(lo, 16384.f, MAGIC), MAGIC - 32768.f, MAGIC + 32767.f);
  const float b = __builtin_amdgcn_fmed3f(fmaf(hi, 16384.f, MAGIC), MAGIC - 32768.f, MAGIC + 32767.f);
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x05040100u);   // b.lo16 : a.lo16
}
__device__ __forceinline__ void unpack_q14(unsigned w, float& lo, float& hi) {
  lo = (float)(int)(short)(w & 0xffffu) * (1.0f / 16384.f);
  hi = (float)((int)w >> 16) * (1.0f / 16384.f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- two-stage sums (instead of float atomics): every producer workgroup stores its partial result in its own slab of a
// caller-provided workspace, and this kernel adds the slabs in a FIXED order -- bitwise reproducible from launch to launch and
// independent of how the workgroups were scheduled.  out[e] (+)= scale * sum_k parts[k * stride + e].
// TPE = 1: one thread per element, parts in order (few parts, many elements); TPE = 32: 32 lanes per element, each sums
// parts j, j + 32, ... in order, then a fixed shuffle tree (many parts, few elements).
template <int TPE>
static __global__ __launch_bounds__(256) void o2_sum_parts_kernel(const float* __restrict__ parts, int nparts, int64_t stride,
                                                                  float* __restrict__ out, int64_t nelem, float scale,
                                                                  int accumulate) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) / TPE;
  const int j = threadIdx.x % TPE;
  float s = 0.f;
  if (e < nelem)
    for (int k = j; k < nparts; k += TPE) s += parts[(size_t)k * stride + e];
  if (TPE > 1) {
#pragma unroll
    for (int o = TPE / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  }
  if (e < nelem && j == 0) out[e] = accumulate ? out[e] + s * scale : s * scale;
}
static inline void o2_sum_parts(const float* parts, int nparts, int64_t stride, float* out, int64_t nelem, float scale,
                                int accumulate, hipStream_t s) {
  if (nparts > 64) {
    hipLaunchKernelGGL(o2_sum_parts_kernel<32>, dim3((unsigned)((nelem * 32 + 255) / 256)), dim3(256), 0, s, parts, nparts,
                       stride, out, nelem, scale, accumulate);
  } else {
    hipLaunchKernelGGL(o2_sum_parts_kernel<1>, dim3((unsigned)((nelem + 255) / 256)), dim3(256), 0, s, parts, nparts, stride,
                       out, nelem, scale, accumulate);
  }
}

#define O2_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return O2_ERR_LAUNCH;            \
  } while (0)
