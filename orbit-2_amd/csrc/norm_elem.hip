// LayerNorm fwd/bwd + HBM-bound elementwise / reduction kernels (bf16 rows, 16-byte accesses).
#include "common.h"
#include "../../include/orbit2_hip.h"

namespace {

constexpr int LN_MAXD = 8192;  // up to 16 16-byte chunks per lane

__device__ __forceinline__ void unpack8(const u32x4 r, float* f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f((bf16_t)(r[j] & 0xffff)); f[2 * j + 1] = bf2f((bf16_t)(r[j] >> 16)); }
}

// one wave per row; 4 rows per 256-thread block; NC = 16-byte chunks per lane (compile time)
template <int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma,
                                                     const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int D, float eps, int ldy) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = D >> 3;
  const bf16_t* xr = x + (size_t)row * D;
  float v[NC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + c * 64;
    if (ch < nch) {
      unpack8(*reinterpret_cast<const u32x4*>(xr + ch * 8), v[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[c][j];
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + c * 64;
    if (ch < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mu; q += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  bf16_t* yr = y + (size_t)row * ldy;      // ldy: row pitch of y (>= D)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + c * 64;
    if (ch < nch) {
      float g[8], b[8];
      unpack8(*reinterpret_cast<const u32x4*>(gamma + ch * 8), g);
      unpack8(*reinterpret_cast<const u32x4*>(beta + ch * 8), b);
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack_bf2((v[c][2 * j] - mu) * rs * g[2 * j] + b[2 * j],
                        (v[c][2 * j + 1] - mu) * rs * g[2 * j + 1] + b[2 * j + 1]);
      *reinterpret_cast<u32x4*>(yr + ch * 8) = o;
    }
  }
}

// Each wave walks LN_RPW consecutive rows (two passes per row: statistics, then dx from the L1/L2-hot
// lines) and keeps its dgamma/dbeta partial sums in registers; one partial row per block (D % 4 == 0).
constexpr int LN_RPW = 16;
#ifndef O2_LN_FMA
#define O2_LN_FMA 1      // row-shared backward: the fused-multiply-add form of xhat and of dx (0: the expressions of rounds 2-5, A/B builds)
#endif
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                     const bf16_t* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                     bf16_t* __restrict__ dx, float* __restrict__ part, int rows,
                                                     int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 3;
  const int pw = blockIdx.x * 4 + wave;
  float dg[NC][8], db[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[c][j] = 0.f; db[c][j] = 0.f; }
  // gamma is row-invariant: unpacked once per wave would cost 48 VGPRs; it stays packed (6 x 16 B per lane)
  u32x4 gp[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + c * 64;
    gp[c] = ch < nch ? *reinterpret_cast<const u32x4*>(gamma + ch * 8) : (u32x4){0u, 0u, 0u, 0u};
  }
  for (int rr = 0; rr < LN_RPW; ++rr) {
    const int row = pw * LN_RPW + rr;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    const bf16_t* xr = x + (size_t)row * D;
    const bf16_t* dyr = dy + (size_t)row * D;
    // every global load of the row is issued here, up front (x, dy and the residual-stream gradient), and the row
    // stays PACKED in registers between the statistics pass and the dx pass: one HBM round trip per row instead
    // of a second trip through L1/L2 plus a late dres fetch
    u32x4 xp[NC], dp[NC], rp[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
        xp[c] = *reinterpret_cast<const u32x4*>(xr + ch * 8);
        dp[c] = *reinterpret_cast<const u32x4*>(dyr + ch * 8);
        if (dres) rp[c] = *reinterpret_cast<const u32x4*>(dres + (size_t)row * D + ch * 8);
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
        float xv[8], dv[8], gv[8];
        unpack8(xp[c], xv);
        unpack8(dp[c], dv);
        unpack8(gp[c], gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xhat = (xv[j] - mu) * rs;
          const float g = dv[j] * gv[j];
          s1 += g; s2 += g * xhat;
          dg[c][j] += dv[j] * xhat; db[c][j] += dv[j];
        }
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    bf16_t* dxr = dx + (size_t)row * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
        float xv[8], dv[8], gv[8], o[8];
        unpack8(xp[c], xv);
        unpack8(dp[c], dv);
        unpack8(gp[c], gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (dv[j] * gv[j] - s1 - (xv[j] - mu) * rs * s2);
        if (dres) {
          float rv[8];
          unpack8(rp[c], rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += rv[j];
        }
        u32x4 ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = pack_bf2(o[2 * j], o[2 * j + 1]);
        *reinterpret_cast<u32x4*>(dxr + ch * 8) = ov;
      }
    }
  }
  // the four waves' partial sums are combined in LDS (wave 0 stores, waves 1..3 add in turn): one partial row per
  // block instead of four -> the reduce kernel reads a quarter of the data
  extern __shared__ float ln_red[];            // [2][D]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float* a = ln_red + ch * 8 + j;
            float* b = ln_red + D + ch * 8 + j;
            *a = (w ? *a : 0.f) + dg[c][j];
            *b = (w ? *b : 0.f) + db[c][j];
          }
        }
      }
    }
    __syncthreads();
  }
  float* pout = part + (size_t)blockIdx.x * 2 * D;
  for (int e = threadIdx.x * 4; e < 2 * D; e += 256 * 4)
    *reinterpret_cast<f32x4*>(pout + e) = *reinterpret_cast<const f32x4*>(ln_red + e);
}

// Row-shared form.  Wide rows (D > 4096, interm_10b: D = 8192): the per-wave form above would keep 2 x 8 x 16 gradient
// accumulators plus the packed row in registers and spill.  Here the TPB/64 waves of the block share every row, each
// thread owning the chunks tid, tid + TPB, ..: a quarter of the accumulators per thread, one LDS exchange + barrier per row for the two
// row statistics (double-buffered slots), the next row's loads issued before that barrier, and the block's partial
// dgamma / dbeta row written straight from registers (the waves own disjoint columns).
// EXACT: D == 8 NCW TPB -- no chunk is ever out of range, so no lane is ever masked off: without the tests the loop loses its
// exec-mask bookkeeping (19 s_and_saveexec + 20 s_or + the copies that merge the masked paths)
template <int NCW, int TPB = 256, int RPB = 4 * LN_RPW, bool EXACT = false>
__global__ __launch_bounds__(TPB) void ln_bwd_wide_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ gamma,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                          float* __restrict__ part, int rows, int D) {
  __shared__ float red[2][TPB / 64][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = D >> 3;
  const int row0 = blockIdx.x * RPB;
  const int nrow = (rows - row0) < RPB ? (rows - row0) : RPB;
  float dg[NCW][8], db[NCW][8];
  u32x4 gp[NCW];
#pragma unroll
  for (int c = 0; c < NCW; ++c) {
    const int ch = tid + c * TPB;
    gp[c] = (EXACT || ch < nch) ? *reinterpret_cast<const u32x4*>(gamma + ch * 8) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[c][j] = 0.f; db[c][j] = 0.f; }
  }
  u32x4 xp[NCW], dp[NCW], rp[NCW], xn[NCW], dn[NCW], rn[NCW];
  auto load_row = [&](int row, u32x4* xo, u32x4* dyo, u32x4* ro) {
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
      const int ch = tid + c * TPB;
      if (EXACT || ch < nch) {
        xo[c] = *reinterpret_cast<const u32x4*>(x + (size_t)row * D + ch * 8);
        dyo[c] = *reinterpret_cast<const u32x4*>(dy + (size_t)row * D + ch * 8);
        if (dres) ro[c] = *reinterpret_cast<const u32x4*>(dres + (size_t)row * D + ch * 8);
      }
    }
  };
  if (nrow > 0) load_row(row0, xp, dp, rp);
  for (int rr = 0; rr < nrow; ++rr) {
    const int row = row0 + rr;
    const float mu = mean[row], rs = rstd[row];
    const float nmurs = -mu * rs;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
      if (EXACT || tid + c * TPB < nch) {
        float xv[8], dv[8], gv[8];
        unpack8(xp[c], xv); unpack8(dp[c], dv); unpack8(gp[c], gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xhat = O2_LN_FMA ? fmaf(xv[j], rs, nmurs) : (xv[j] - mu) * rs;     // (x - mu) rs as ONE fma
          const float g = dv[j] * gv[j];
          s1 += g; s2 += g * xhat;
          dg[c][j] += dv[j] * xhat; db[c][j] += dv[j];
        }
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) { red[rr & 1][wave][0] = s1; red[rr & 1][wave][1] = s2; }
    if (rr + 1 < nrow) load_row(row + 1, xn, dn, rn);      // in flight across the barrier
    __syncthreads();
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) { s1 += red[rr & 1][w][0]; s2 += red[rr & 1][w][1]; }
    s1 /= (float)D; s2 /= (float)D;
    // dx = rs (dy g - s1 - xhat s2) = dy (g rs) + x c2 + c0 with the row's c2 = -rs^2 s2, c0 = rs (mu rs s2 - s1): three
    // instructions per element (one multiply, two fused multiply-adds) instead of seven -- the kernel is co-limited by vector issue
    const float c2 = -rs * rs * s2, c0 = rs * (mu * rs * s2 - s1);
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
      const int ch = tid + c * TPB;
      if (EXACT || ch < nch) {
        float xv[8], dv[8], gv[8], o[8];
        unpack8(xp[c], xv); unpack8(dp[c], dv); unpack8(gp[c], gv);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = O2_LN_FMA ? fmaf(xv[j], c2, fmaf(dv[j], gv[j] * rs, c0)) : rs * (dv[j] * gv[j] - s1 - (xv[j] - mu) * rs * s2);
        if (dres) {
          float rv[8];
          unpack8(rp[c], rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += rv[j];
        }
        u32x4 ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = pack_bf2(o[2 * j], o[2 * j + 1]);
        *reinterpret_cast<u32x4*>(dx + (size_t)row * D + ch * 8) = ov;
      }
    }
#pragma unroll
    for (int c = 0; c < NCW; ++c) { xp[c] = xn[c]; dp[c] = dn[c]; rp[c] = rn[c]; }
  }
  float* pout = part + (size_t)blockIdx.x * 2 * D;
#pragma unroll
  for (int c = 0; c < NCW; ++c) {
    const int ch = tid + c * TPB;
    if (EXACT || ch < nch) {
      *reinterpret_cast<f32x4*>(pout + ch * 8) = (f32x4){dg[c][0], dg[c][1], dg[c][2], dg[c][3]};
      *reinterpret_cast<f32x4*>(pout + ch * 8 + 4) = (f32x4){dg[c][4], dg[c][5], dg[c][6], dg[c][7]};
      *reinterpret_cast<f32x4*>(pout + D + ch * 8) = (f32x4){db[c][0], db[c][1], db[c][2], db[c][3]};
      *reinterpret_cast<f32x4*>(pout + D + ch * 8 + 4) = (f32x4){db[c][4], db[c][5], db[c][6], db[c][7]};
    }
  }
}

// out[which][col] = beta*out + sum_p part[p][which][col]; 32 columns x 8 partial-groups per block
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int D,
                                                            void* dgamma, void* dbeta, int fp32, float beta) {
  __shared__ float red[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + tx;          // flat (which, col) index in [0, 2D)
  float s = 0.f;
  if (e < 2 * D) {
    const int which = e / D, col = e - which * D;
    // eight partial rows in flight per thread (one dependent add chain per row kept the launch at 0.5 TB/s: 100 us for
    // 50 MB at the interm_1b shape); fixed order: the eight chains, then their tree
    const float* src = part + (size_t)which * D + col;
    const size_t step = (size_t)2 * D;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int p = ty;
    for (; p + 56 < nblk; p += 64) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += src[(size_t)(p + 8 * k) * step];
    }
    for (int k = 0; p < nblk; p += 8, ++k) a[k] += src[(size_t)p * step];
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && e < 2 * D) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][tx];
    const int which = e / D, col = e - which * D;
    void* dst = which ? dbeta : dgamma;
    if (fp32) {
      float* o = (float*)dst + col;
      *o = t + (beta != 0.f ? beta * *o : 0.f);
    } else {
      bf16_t* o = (bf16_t*)dst + col;
      *o = f2bf(t + (beta != 0.f ? beta * bf2f(*o) : 0.f));
    }
  }
}

// ---- what a GEMM epilogue does AFTER a cross-rank reduction of partial products -----------
// y = residual + rowscale[m / rps] * dropout(x + addend[m % res_mod])   (every term optional; same mask hash and
// order of operations as gemm.hip's epilogue, so a tensor-parallel block equals the single-rank fused one)
__global__ __launch_bounds__(256) void post_reduce_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ addend,
                                                          int res_mod, const bf16_t* __restrict__ residual,
                                                          bf16_t* __restrict__ y, int M, int N, unsigned thr,
                                                          float dscale, uint64_t seed,
                                                          const float* __restrict__ rowscale, int rows_per_scale) {
  const int64_t nch = (int64_t)M * N / 8;
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < nch; ch += (int64_t)gridDim.x * 256) {
    const int64_t idx = ch * 8;
    const int m = (int)(idx / N);
    const int n = (int)(idx - (int64_t)m * N);
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + idx);
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f((bf16_t)(v[j] & 0xffff)); f[2 * j + 1] = bf2f((bf16_t)(v[j] >> 16)); }
    if (addend) {
      const int rm = res_mod > 0 ? (m % res_mod) : m;
      const u32x4 a = *reinterpret_cast<const u32x4*>(addend + (size_t)rm * N + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[2 * j] += bf2f((bf16_t)(a[j] & 0xffff)); f[2 * j + 1] += bf2f((bf16_t)(a[j] >> 16)); }
    }
    if (thr) {
      const uint64_t sd = seed ^ o2_seed_salt;
      const uint32_t h0 = o2_hash64(sd, (uint64_t)idx >> 2), h1 = o2_hash64(sd, ((uint64_t)idx >> 2) + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[j] = (((h0 >> (8 * j)) & 0xffu) >= thr) ? f[j] * dscale : 0.f;
        f[4 + j] = (((h1 >> (8 * j)) & 0xffu) >= thr) ? f[4 + j] * dscale : 0.f;
      }
    }
    if (rowscale) {
      const float s = rowscale[m / rows_per_scale];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] *= s;
    }
    if (residual) {
      const u32x4 r = *reinterpret_cast<const u32x4*>(residual + idx);
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[2 * j] += bf2f((bf16_t)(r[j] & 0xffff)); f[2 * j + 1] += bf2f((bf16_t)(r[j] >> 16)); }
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf2(f[2 * j], f[2 * j + 1]);
    *reinterpret_cast<u32x4*>(y + idx) = o;
  }
}

// ---- dropout backward ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dym,
                                                          int M, int N, unsigned thr, float dscale, uint64_t seed,
                                                          const float* __restrict__ rowscale, int rows_per_scale) {
  const int64_t nch = (int64_t)M * N / 8;
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < nch; ch += (int64_t)gridDim.x * 256) {
    const int64_t idx = ch * 8;
    const int m = (int)(idx / N);
    const u32x4 v = *reinterpret_cast<const u32x4*>(dy + idx);
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f((bf16_t)(v[j] & 0xffff)); f[2 * j + 1] = bf2f((bf16_t)(v[j] >> 16)); }
    if (thr) {
      const uint64_t sd = seed ^ o2_seed_salt;
      const uint32_t h0 = o2_hash64(sd, (uint64_t)idx >> 2), h1 = o2_hash64(sd, ((uint64_t)idx >> 2) + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[j] = (((h0 >> (8 * j)) & 0xffu) >= thr) ? f[j] * dscale : 0.f;
        f[4 + j] = (((h1 >> (8 * j)) & 0xffu) >= thr) ? f[4 + j] * dscale : 0.f;
      }
    }
    if (rowscale) {
      const float s = rowscale[m / rows_per_scale];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] *= s;
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf2(f[2 * j], f[2 * j + 1]);
    *reinterpret_cast<u32x4*>(dym + idx) = o;
  }
}

// ---- column sums: stage 1 partials [P][N] (16-byte loads, 8 row-lanes per block), stage 2 reduce ----------
// rows per partial sum: 128 for the large problems; 32 when there are few rows (the 32 x 64-grid presets: 4096 rows x 1024 columns
// were 128 workgroups walking 16 rows each one after the other -- half the CUs, latency-bound; 512 workgroups of 4 rows now).
// One function decides for the workspace query and both launchers.
static inline int cs_rows(int M) { return M >= 32768 ? 128 : 32; }
template <bool FP32>
__global__ __launch_bounds__(256) void colsum_part_kernel(const void* __restrict__ xv, int M, int N, int ldx,
                                                          float* __restrict__ part, int rpp) {
  __shared__ float red[8][32][9];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = (blockIdx.x * 32 + tx) * 8;
  const int r0 = blockIdx.y * rpp;
  const int r1 = r0 + rpp < M ? r0 + rpp : M;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    if (FP32) {
      const float* x = (const float*)xv;
      for (int r = r0 + ty; r < r1; r += 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + c0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + c0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] += a[j]; acc[4 + j] += b[j]; }
      }
    } else {
      const bf16_t* x = (const bf16_t*)xv;
      for (int r = r0 + ty; r < r1; r += 8) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + (size_t)r * ldx + c0), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += f[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx][j] = acc[j];
  __syncthreads();
  if (ty == 0 && c0 < N) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k][tx][j];
      part[(size_t)blockIdx.y * N + c0 + j] = t;
    }
  }
}
// dropout / DropPath backward FUSED with the column sums of its result (the bias gradient of the Linear whose output
// gradient this is): the same [128 rows x 256 columns] blocks, thread mapping and summation order as colsum_part_kernel, so
// the partials -- and the bias gradient -- are bit-identical to dropout_bwd + colsum, without re-reading dym from HBM.
__global__ __launch_bounds__(256) void dropout_bwd_colsum_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dym,
                                                                 int M, int N, unsigned thr, float dscale, uint64_t seed,
                                                                 const float* __restrict__ rowscale, int rows_per_scale,
                                                                 float* __restrict__ part, int rpp) {
  __shared__ float red[8][32][9];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = (blockIdx.x * 32 + tx) * 8;
  const int r0 = blockIdx.y * rpp;
  const int r1 = r0 + rpp < M ? r0 + rpp : M;
  const uint64_t sd = seed ^ o2_seed_salt;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    for (int r = r0 + ty; r < r1; r += 8) {
      const int64_t idx = (int64_t)r * N + c0;
      float f[8];
      unpack8(*reinterpret_cast<const u32x4*>(dy + idx), f);
      if (thr) {
        const uint32_t h0 = o2_hash64(sd, (uint64_t)idx >> 2), h1 = o2_hash64(sd, ((uint64_t)idx >> 2) + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f[j] = (((h0 >> (8 * j)) & 0xffu) >= thr) ? f[j] * dscale : 0.f;
          f[4 + j] = (((h1 >> (8 * j)) & 0xffu) >= thr) ? f[4 + j] * dscale : 0.f;
        }
      }
      if (rowscale) {
        const float sc = rowscale[r / rows_per_scale];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= sc;
      }
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack_bf2(f[2 * j], f[2 * j + 1]);
      *reinterpret_cast<u32x4*>(dym + idx) = o;
      unpack8(o, f);                                   // the sum is over the ROUNDED values (what colsum would read back)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx][j] = acc[j];
  __syncthreads();
  if (ty == 0 && c0 < N) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k][tx][j];
      part[(size_t)blockIdx.y * N + c0 + j] = t;
    }
  }
}
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int P, int N, void* out,
                                                            int out_fp32, float beta) {
  __shared__ float red[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;
  float s = 0.f;
  if (col < N) {
    const float* src = part + col;           // eight partial rows in flight per thread, fixed order (see ln_bwd_reduce_kernel)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int p = ty;
    for (; p + 56 < P; p += 64) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += src[(size_t)(p + 8 * k) * N];
    }
    for (int k = 0; p < P; p += 8, ++k) a[k] += src[(size_t)p * N];
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][tx];
    if (out_fp32) {
      float* o = (float*)out + col;
      *o = t + (beta != 0.f ? beta * *o : 0.f);
    } else {
      bf16_t* o = (bf16_t*)out + col;
      *o = f2bf(t + (beta != 0.f ? beta * bf2f(*o) : 0.f));
    }
  }
}

__global__ __launch_bounds__(256) void batch_sum_kernel(const bf16_t* __restrict__ x, void* out, int B, int64_t n,
                                                        int out_fp32, float beta) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += bf2f(x[(size_t)b * n + i]);
    if (out_fp32) {
      float* o = (float*)out + i;
      *o = s + (beta != 0.f ? beta * *o : 0.f);
    } else {
      bf16_t* o = (bf16_t*)out + i;
      *o = f2bf(s + (beta != 0.f ? beta * bf2f(*o) : 0.f));
    }
  }
}

__global__ __launch_bounds__(256) void cast_f2b_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(s + i * 4);
    u32x2 o; o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(d + i * 4) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) d[n4 * 4 + threadIdx.x] = f2bf(s[n4 * 4 + threadIdx.x]);
}
__global__ __launch_bounds__(256) void cast_b2f_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = bf2f(s[i]);
}
__global__ __launch_bounds__(256) void add_rowvec_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ vec,
                                                         bf16_t* __restrict__ y, int rows, int N) {
  const int64_t n = (int64_t)rows * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = f2bf(bf2f(a[i]) + bf2f(vec[i % N]));
}

// ---- bf16 matrix transpose through LDS: dst[C][R] = src[R][C]  (transposed compute copies of the weights so
//      that the input-gradient GEMM dX = dY . W runs in the K-contiguous form) ---------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             int R, int C) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R) dst[(size_t)c * R + r] = tile[tx][i];
  }
}

// ---- AdamW ---------------------------------------------------------------------------------
template <bool GFP32>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                    const void* __restrict__ gv, bf16_t* __restrict__ p16, int64_t n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float rsbc2, float gs, const float* __restrict__ found_inf) {
  if (found_inf && *found_inf != 0.f) return;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    float g[4], pp[4], mm[4], vv[4];
    const int cnt = (n - i) >= 4 ? 4 : (int)(n - i);
    if (cnt == 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + i), b = *reinterpret_cast<const f32x4*>(m + i),
                  c = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) { pp[j] = a[j]; mm[j] = b[j]; vv[j] = c[j]; }
      if (GFP32) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>((const float*)gv + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = gg[j];
      } else {
        const u32x2 gg = *reinterpret_cast<const u32x2*>((const bf16_t*)gv + i);
        g[0] = bf2f((bf16_t)(gg[0] & 0xffff)); g[1] = bf2f((bf16_t)(gg[0] >> 16));
        g[2] = bf2f((bf16_t)(gg[1] & 0xffff)); g[3] = bf2f((bf16_t)(gg[1] >> 16));
      }
    } else {
      for (int j = 0; j < cnt; ++j) {
        pp[j] = p[i + j]; mm[j] = m[i + j]; vv[j] = v[i + j];
        g[j] = GFP32 ? ((const float*)gv)[i + j] : bf2f(((const bf16_t*)gv)[i + j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < cnt) {
        const float gj = g[j] * gs;
        pp[j] *= (1.f - lr * wd);
        mm[j] = b1 * mm[j] + (1.f - b1) * gj;
        vv[j] = b2 * vv[j] + (1.f - b2) * gj * gj;
        const float denom = sqrtf(vv[j]) * rsbc2 + eps;
        pp[j] -= (lr / bc1) * (mm[j] / denom);
      }
    }
    if (cnt == 4) {
      *reinterpret_cast<f32x4*>(p + i) = (f32x4){pp[0], pp[1], pp[2], pp[3]};
      *reinterpret_cast<f32x4*>(m + i) = (f32x4){mm[0], mm[1], mm[2], mm[3]};
      *reinterpret_cast<f32x4*>(v + i) = (f32x4){vv[0], vv[1], vv[2], vv[3]};
      if (p16) {
        u32x2 o; o[0] = pack_bf2(pp[0], pp[1]); o[1] = pack_bf2(pp[2], pp[3]);
        *reinterpret_cast<u32x2*>(p16 + i) = o;
      }
    } else {
      for (int j = 0; j < cnt; ++j) {
        p[i + j] = pp[j]; m[i + j] = mm[j]; v[i + j] = vv[j];
        if (p16) p16[i + j] = f2bf(pp[j]);
      }
    }
  }
}

template <bool GFP32>
__global__ __launch_bounds__(256) void check_finite_kernel(const void* __restrict__ gv, int64_t n, float* found) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float g = GFP32 ? ((const float*)gv)[i] : bf2f(((const bf16_t*)gv)[i]);
    bad |= !(fabsf(g) <= 3.0e38f);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *found = 1.f;
}

// per-sample DropPath scales (timm DropPath semantics: Bernoulli(keep)/keep per sample), 32-bit threshold
__global__ void droppath_scales_kernel(float* __restrict__ out, int B, float p, uint64_t seed) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const uint32_t h = o2_hash64(seed ^ o2_seed_salt ^ 0xD1B54A32D192ED03ull, (uint64_t)b);
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  out[b] = (u >= p) ? 1.0f / (1.0f - p) : 0.f;
}

inline int grid_for(int64_t work_items, int per_block) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (int)g;
}

}  // namespace

#define LN_DISPATCH(NCV, CALL)            \
  do {                                    \
    if (NCV <= 1) { CALL(1); }            \
    else if (NCV <= 2) { CALL(2); }       \
    else if (NCV <= 4) { CALL(4); }       \
    else if (NCV <= 6) { CALL(6); }       \
    else if (NCV <= 8) { CALL(8); }       \
    else { CALL(16); }                    \
  } while (0)

extern "C" int orbit2_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean,
                                    float* rstd, int rows, int D, float eps, void* stream) {
  return orbit2_layernorm_fwd_ld(x, gamma, beta, y, mean, rstd, rows, D, D, eps, stream);
}

extern "C" int orbit2_layernorm_fwd_ld(const void* x, const void* gamma, const void* beta, void* y, float* mean,
                                       float* rstd, int rows, int D, int ldy, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || D <= 0 || (D & 7) || D > LN_MAXD || ldy < D || (ldy & 7))
    return O2_ERR_ARG;
  const int nc = (D / 8 + 63) / 64;
#define CALL(N)                                                                                              \
  hipLaunchKernelGGL(ln_fwd_kernel<N>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,              \
                     (const bf16_t*)x, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, mean, rstd,  \
                     rows, D, eps, ldy)
  LN_DISPATCH(nc, CALL);
#undef CALL
  O2_CHECK_LAUNCH();
  return O2_OK;
}

static inline int ln_bwd_nparts(int rows) { return ((rows + 4 * LN_RPW - 1) / (4 * LN_RPW)) * 4; }
// few rows (the 32 x 64-grid presets): LN_FEW_RPB rows per workgroup of the row-shared form -- 4096 rows at 16 per workgroup were
// 256 workgroups of two waves walking their rows one global-load latency after the other (18 us at D = 1024, a quarter of it
// bandwidth); 4 per workgroup put four workgroups on every CU
constexpr int LN_FEW_RPB = 4;
static inline int ln_bwd_nparts_few(int rows) { return (rows + LN_FEW_RPB - 1) / LN_FEW_RPB; }
extern "C" int orbit2_layernorm_bwd_ws_floats(int rows, int D) {
  const int a = ln_bwd_nparts(rows), b = rows < 32768 ? ln_bwd_nparts_few(rows) : 0;
  return (a > b ? a : b) * 2 * D;
}

extern "C" int orbit2_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean,
                                    const float* rstd, const void* dres, void* dx, void* dgamma, void* dbeta,
                                    int grads_fp32, float beta_acc, float* ws, int ws_floats, int rows, int D,
                                    void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !ws) return O2_ERR_ARG;
  if (rows <= 0 || (D & 7) || D > LN_MAXD) return O2_ERR_ARG;
  const int nparts = ln_bwd_nparts(rows);
  if (ws_floats < orbit2_layernorm_bwd_ws_floats(rows, D)) return O2_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nc = (D / 8 + 63) / 64;
  // row-shared form for the model widths (8m / 117m / 1b / 10b); 64 rows per block, LN_FEW_RPB when there are few rows (the
  // small grids: 4096 rows would otherwise occupy 64 CUs).  The workspace holds 4 partial rows per 64 input rows.
  const bool few = rows < 32768;
  const int npart_used = few ? ln_bwd_nparts_few(rows) : nparts / 4;
#define WIDE_(NCW, TPB, EX)                                                                                         \
  do {                                                                                                              \
    if (few)                                                                                                        \
      hipLaunchKernelGGL((ln_bwd_wide_kernel<NCW, TPB, LN_FEW_RPB, EX>), dim3(npart_used), dim3(TPB), 0, s, (const bf16_t*)dy, \
                         (const bf16_t*)x, (const bf16_t*)gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, ws,  \
                         rows, D);                                                                                  \
    else                                                                                                            \
      hipLaunchKernelGGL((ln_bwd_wide_kernel<NCW, TPB, 64, EX>), dim3(npart_used), dim3(TPB), 0, s, (const bf16_t*)dy,  \
                         (const bf16_t*)x, (const bf16_t*)gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, ws,  \
                         rows, D);                                                                                  \
  } while (0)
#define WIDE(NCW, TPB)                                                                                              \
  do {                                                                                                              \
    if (D == 8 * (NCW) * (TPB)) WIDE_(NCW, TPB, true);                                                              \
    else WIDE_(NCW, TPB, false);                                                                                    \
  } while (0)
  if (D == 3072) WIDE(3, 128);          // 363 vs 414 us (per-wave form) at 65536 rows
  else if (D == 1024) WIDE(1, 128);
  else if (D == 256) WIDE(1, 64);
  else if (nc > 8) WIDE(4, 256);        // D in (4096, 8192]: the per-wave form spills (1127 -> 115 us at D = 8192)
  else if (few) {
    // other widths with few rows: per-wave form, its partition (64 rows per block)
#define CALL(N)                                                                                                  \
  hipLaunchKernelGGL(ln_bwd_kernel<N>, dim3(nparts / 4), dim3(256), 2 * D * sizeof(float), s, (const bf16_t*)dy, (const bf16_t*)x, \
                     (const bf16_t*)gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, ws, rows, D)
    LN_DISPATCH(nc, CALL);
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * D + 31) / 32), dim3(256), 0, s, ws, nparts / 4, D, dgamma, dbeta,
                       grads_fp32, beta_acc);
    O2_CHECK_LAUNCH();
    return O2_OK;
  } else {
    LN_DISPATCH(nc, CALL);
#undef CALL
  }
#undef WIDE
#undef WIDE_
  O2_CHECK_LAUNCH();
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * D + 31) / 32), dim3(256), 0, s, ws, npart_used, D, dgamma, dbeta,
                     grads_fp32, beta_acc);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_dropout_bwd(const void* dy, void* dym, int M, int N, float drop_p, uint64_t seed,
                                  const float* rowscale, int rows_per_scale, void* stream) {
  if (!dy || !dym || M <= 0 || N <= 0 || (N & 7) || drop_p < 0.f || drop_p >= 1.f) return O2_ERR_ARG;
  if (rowscale && rows_per_scale <= 0) return O2_ERR_ARG;
  const unsigned thr = (unsigned)(drop_p * 256.0f + 0.5f);
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for((int64_t)M * N / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (bf16_t*)dym, M, N, thr, 256.0f / (256.0f - (float)thr), seed, rowscale,
                     rows_per_scale);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_post_reduce(const void* x, const void* addend, int res_mod, const void* residual, void* y, int M,
                                  int N, float drop_p, uint64_t seed, const float* rowscale, int rows_per_scale,
                                  void* stream) {
  if (!x || !y || M <= 0 || N <= 0 || (N & 7) || drop_p < 0.f || drop_p >= 1.f || res_mod < 0) return O2_ERR_ARG;
  if (rowscale && rows_per_scale <= 0) return O2_ERR_ARG;
  const unsigned thr = (unsigned)(drop_p * 256.0f + 0.5f);
  hipLaunchKernelGGL(post_reduce_kernel, dim3(grid_for((int64_t)M * N / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)addend, res_mod, (const bf16_t*)residual, (bf16_t*)y, M, N, thr,
                     256.0f / (256.0f - (float)thr), seed, rowscale, rows_per_scale);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_colsum_ws_floats(int M, int N) { return ((M + cs_rows(M) - 1) / cs_rows(M)) * N; }

extern "C" int orbit2_colsum(const void* x, int x_fp32, int M, int N, int ldx, void* out, int out_fp32, float beta,
                             float* ws, int ws_floats, void* stream) {
  if (!x || !out || !ws || M <= 0 || N <= 0) return O2_ERR_ARG;
  const int rpp = cs_rows(M), P = (M + rpp - 1) / rpp;
  if (ws_floats < P * N) return O2_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if ((N & 7) || (ldx & 7)) return O2_ERR_ARG;
  dim3 g1((N + 255) / 256, P);
  if (x_fp32) hipLaunchKernelGGL(colsum_part_kernel<true>, g1, dim3(256), 0, s, x, M, N, ldx, ws, rpp);
  else hipLaunchKernelGGL(colsum_part_kernel<false>, g1, dim3(256), 0, s, x, M, N, ldx, ws, rpp);
  O2_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 31) / 32), dim3(256), 0, s, ws, P, N, out, out_fp32, beta);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_dropout_bwd_colsum(const void* dy, void* dym, int M, int N, float drop_p, uint64_t seed,
                                         const float* rowscale, int rows_per_scale, void* colsum_out, int out_fp32, float beta,
                                         float* ws, int ws_floats, void* stream) {
  if (!dy || !dym || !colsum_out || !ws || M <= 0 || N <= 0 || (N & 7)) return O2_ERR_ARG;
  if (drop_p < 0.f || drop_p >= 1.f || (rowscale && rows_per_scale <= 0)) return O2_ERR_ARG;
  const int rpp = cs_rows(M), P = (M + rpp - 1) / rpp;
  if (ws_floats < P * N) return O2_ERR_ARG;
  const unsigned thr = (unsigned)(drop_p * 256.0f + 0.5f);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dropout_bwd_colsum_kernel, dim3((N + 255) / 256, P), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dym, M, N,
                     thr, 256.0f / (256.0f - (float)thr), seed, rowscale, rows_per_scale, ws, rpp);
  O2_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 31) / 32), dim3(256), 0, s, ws, P, N, colsum_out, out_fp32, beta);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_batch_sum(const void* x, void* out, int B, int rows, int N, int out_fp32, float beta,
                                void* stream) {
  if (!x || !out || B <= 0 || rows <= 0 || N <= 0) return O2_ERR_ARG;
  const int64_t n = (int64_t)rows * N;
  hipLaunchKernelGGL(batch_sum_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     out, B, n, out_fp32, beta);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return O2_ERR_ARG;
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, n);
  O2_CHECK_LAUNCH();
  return O2_OK;
}
extern "C" int orbit2_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return O2_ERR_ARG;
  hipLaunchKernelGGL(cast_b2f_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     dst, n);
  O2_CHECK_LAUNCH();
  return O2_OK;
}
extern "C" int orbit2_add_rowvec(const void* a, const void* vec, void* y, int rows, int N, void* stream) {
  if (!a || !vec || !y || rows <= 0 || N <= 0) return O2_ERR_ARG;
  hipLaunchKernelGGL(add_rowvec_kernel, dim3(grid_for((int64_t)rows * N, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)a, (const bf16_t*)vec, (bf16_t*)y, rows, N);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_adamw(float* p, float* m, float* v, const void* g, int g_fp32, void* p16, int64_t n, float lr,
                            float beta1, float beta2, float eps, float wd, float bc1, float bc2, float grad_scale,
                            const float* found_inf, void* stream) {
  if (!p || !m || !v || !g || n <= 0) return O2_ERR_ARG;
  if (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) return O2_ERR_ARG;
  if (p16 && ((uintptr_t)p16 & 7)) return O2_ERR_ARG;
  const float rsbc2 = 1.0f / sqrtf(bc2);
  dim3 grid(grid_for((n + 3) / 4, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (g_fp32)
    hipLaunchKernelGGL(adamw_kernel<true>, grid, block, 0, s, p, m, v, g, (bf16_t*)p16, n, lr, beta1, beta2, eps, wd,
                       bc1, rsbc2, grad_scale, found_inf);
  else
    hipLaunchKernelGGL(adamw_kernel<false>, grid, block, 0, s, p, m, v, g, (bf16_t*)p16, n, lr, beta1, beta2, eps, wd,
                       bc1, rsbc2, grad_scale, found_inf);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_check_finite(const void* g, int g_fp32, int64_t n, float* found_inf, void* stream) {
  if (!g || !found_inf || n <= 0) return O2_ERR_ARG;
  dim3 grid(grid_for(n, 256 * 8)), block(256);
  if (g_fp32) hipLaunchKernelGGL(check_finite_kernel<true>, grid, block, 0, (hipStream_t)stream, g, n, found_inf);
  else hipLaunchKernelGGL(check_finite_kernel<false>, grid, block, 0, (hipStream_t)stream, g, n, found_inf);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

O2_DEFINE_SALT_OP(elem)

extern "C" int orbit2_seed_salt(uint64_t value, int add, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  o2_salt_op_gemm(value, add, s);
  o2_salt_op_attn(value, add, s);
  o2_salt_op_elem(value, add, s);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_droppath_scales(float* out, int B, float p, uint64_t seed, void* stream) {
  if (!out || B <= 0 || p < 0.f || p >= 1.f) return O2_ERR_ARG;
  hipLaunchKernelGGL(droppath_scales_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, out, B, p, seed);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_transpose_bf16(const void* src, void* dst, int R, int C, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0) return O2_ERR_ARG;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (bf16_t*)dst, R, C);
  O2_CHECK_LAUNCH();
  return O2_OK;
}
