// Perceptual loss = L1 + 0.5 * mean_b LPIPS-VGG16 (reference: metrics/functional.py:17-33, metrics.py:119-187;
// LPIPS itself is the third-party `lpips` package: ScalingLayer -> torchvision VGG16 features with taps after
// relu1_2/2_2/3_3/4_3/5_3 -> channel unit-normalise -> squared difference -> 1x1 `lin` -> spatial mean -> sum).
//
// Layout: every feature map is NHWC bf16 ([image][y][x][C]), so a 3x3 convolution is a GEMM over
// M = images*H*W pixels, K = 9*Cin, N = Cout: this file supplies the data movement around orbit2_gemm_bf16
// (im2col / col2im with fused ReLU mask and tap-gradient add, 2x2 max-pool), the 3-channel first convolution
// (read straight from the NCHW fp32 prediction / target, ScalingLayer fused), the LPIPS head per tap and the L1 term.
// The reference wraps LPIPS in FSDP MixedPrecision(bf16): bf16 feature maps with fp32 accumulation match that.
#include "common.h"
#include "../../include/orbit2_hip.h"

namespace {

__device__ __forceinline__ void unpack8f(const u32x4& r, float* f) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = bf2f((bf16_t)(r[k] & 0xffff)); f[2 * k + 1] = bf2f((bf16_t)(r[k] >> 16)); }
}
__device__ __forceinline__ u32x4 pack8f(const float* v) {
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
  return o;
}

// ---- im2col: col[p][t][c] = x[p + off_t][c] (zero outside the image), t = ky*3+kx, off = (ky-1, kx-1) ----------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int N,
                                                        int H, int W, int C) {
  const int c8n = C >> 3;
  const int64_t total = (int64_t)N * H * W * 9 * c8n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(e % c8n);
    const int64_t r = e / c8n;
    const int t = (int)(r % 9);
    const int64_t p = r / 9;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (sy >= 0 && sy < H && sx >= 0 && sx < W)
      v = *reinterpret_cast<const u32x4*>(x + ((p + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * C + c8 * 8));
    *reinterpret_cast<u32x4*>(col + e * 8) = v;
  }
}

// ---- col2im: g[p][c] = sum_t dcol[p - off_t][t][c]; optionally dz = (g + tapg) * (act > 0) ------------------------------
__global__ __launch_bounds__(256) void col2im3x3_kernel(const bf16_t* __restrict__ dcol, const bf16_t* __restrict__ act,
                                                        const bf16_t* __restrict__ tapg, bf16_t* __restrict__ out,
                                                        int N, int H, int W, int C) {
  const int c8n = C >> 3;
  const int64_t total = (int64_t)N * H * W * c8n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(e % c8n);
    const int64_t p = e / c8n;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int qy = yy - (t / 3 - 1), qx = xx - (t % 3 - 1);
      if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
        const int64_t q = p - (int64_t)(t / 3 - 1) * W - (t % 3 - 1);
        float f[8];
        unpack8f(*reinterpret_cast<const u32x4*>(dcol + (q * 9 + t) * C + c8 * 8), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += f[k];
      }
    }
    if (act) {
      float a[8];
      unpack8f(*reinterpret_cast<const u32x4*>(act + e * 8), a);
      if (tapg) {
        float g[8];
        unpack8f(*reinterpret_cast<const u32x4*>(tapg + e * 8), g);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += g[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = a[k] > 0.f ? acc[k] : 0.f;
    }
    *reinterpret_cast<u32x4*>(out + e * 8) = pack8f(acc);
  }
}

// ---- 2x2 max-pool (H, W even) ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N,
                                                           int H, int W, int C) {
  const int c8n = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)N * Ho * Wo * c8n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(e % c8n);
    const int64_t po = e / c8n;
    const int xo = (int)(po % Wo), yo = (int)((po / Wo) % Ho);
    const int64_t n = po / ((int64_t)Wo * Ho);
    const bf16_t* src = x + (((n * H + 2 * yo) * W + 2 * xo) * C + c8 * 8);
    float m[8], f[8];
    unpack8f(*reinterpret_cast<const u32x4*>(src), m);
    const int64_t offs[3] = {(int64_t)C, (int64_t)W * C, (int64_t)W * C + C};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      unpack8f(*reinterpret_cast<const u32x4*>(src + offs[j]), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], f[k]);
    }
    *reinterpret_cast<u32x4*>(y + e * 8) = pack8f(m);
  }
}

// dz[pos] = ((pos == first argmax of the window ? g : 0) + tapg[pos]) * (x[pos] > 0)     (x = the pre-pool ReLU output)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ x,
                                                           const bf16_t* __restrict__ tapg, bf16_t* __restrict__ dz,
                                                           int N, int H, int W, int C) {
  const int c8n = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)N * Ho * Wo * c8n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(e % c8n);
    const int64_t po = e / c8n;
    const int xo = (int)(po % Wo), yo = (int)((po / Wo) % Ho);
    const int64_t n = po / ((int64_t)Wo * Ho);
    const int64_t base = ((n * H + 2 * yo) * W + 2 * xo) * C + c8 * 8;
    const int64_t offs[4] = {0, (int64_t)C, (int64_t)W * C, (int64_t)W * C + C};
    float xv[4][8], gv[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) unpack8f(*reinterpret_cast<const u32x4*>(x + base + offs[j]), xv[j]);
    unpack8f(*reinterpret_cast<const u32x4*>(g + e * 8), gv);
    int am[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      am[k] = 0;
      float m = xv[0][k];
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (xv[j][k] > m) { m = xv[j][k]; am[k] = j; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float o[8], tg[8];
      if (tapg) unpack8f(*reinterpret_cast<const u32x4*>(tapg + base + offs[j]), tg);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float v = (am[k] == j) ? gv[k] : 0.f;
        if (tapg) v += tg[k];
        o[k] = xv[j][k] > 0.f ? v : 0.f;
      }
      *reinterpret_cast<u32x4*>(dz + base + offs[j]) = pack8f(o);
    }
  }
}

// ---- first convolution 3 -> 64 on the scaled NCHW fp32 image; w1[(t*3 + ci)][co], t = ky*3+kx -----------------------------
__constant__ float kShift[3] = {-0.030f, -0.088f, -0.188f};
__constant__ float kScale[3] = {0.458f, 0.448f, 0.450f};

__global__ __launch_bounds__(256) void lpips_conv1_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w1,
                                                              const float* __restrict__ b1, bf16_t* __restrict__ out,
                                                              int N, int H, int W) {
  __shared__ float sw[27 * 64 + 64];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) sw[i] = w1[i];
  if (threadIdx.x < 64) sw[27 * 64 + threadIdx.x] = b1[threadIdx.x];
  __syncthreads();
  const int64_t total = (int64_t)N * H * W * 8;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(e & 7);
    const int64_t p = e >> 3;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const int64_t n = p / ((int64_t)W * H);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = sw[27 * 64 + c8 * 8 + k];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
      if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = (img[((n * 3 + ci) * H + sy) * W + sx] - kShift[ci]) / kScale[ci];
        const float* wr = sw + (t * 3 + ci) * 64 + c8 * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fmaf(v, wr[k], acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = fmaxf(acc[k], 0.f);
    *reinterpret_cast<u32x4*>(out + e * 8) = pack8f(acc);
  }
}

// dimg[n][ci][y][x] = (1/scale_ci) * sum_t sum_co dz[p - off_t][co] * w1[t][ci][co]  +  l1_coef * sign(pred - target)
__global__ __launch_bounds__(256) void lpips_conv1_bwd_kernel(const bf16_t* __restrict__ dz, const float* __restrict__ w1,
                                                              const float* __restrict__ pred,
                                                              const float* __restrict__ target, float l1_coef_arg,
                                                              const float* __restrict__ gscale,
                                                              float* __restrict__ dimg, int N, int H, int W) {
  const float l1_coef = gscale ? l1_coef_arg * gscale[0] : l1_coef_arg;     // upstream scalar gradient stays on the device
  __shared__ float sw[27 * 64];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) sw[i] = w1[i];
  __syncthreads();
  const int64_t total = (int64_t)N * H * W;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const int64_t n = p / ((int64_t)W * H);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int t = 0; t < 9; ++t) {
      const int qy = yy - (t / 3 - 1), qx = xx - (t % 3 - 1);
      if (qy < 0 || qy >= H || qx < 0 || qx >= W) continue;
      const bf16_t* src = dz + (p - (int64_t)(t / 3 - 1) * W - (t % 3 - 1)) * 64;
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) {
        float f[8];
        unpack8f(*reinterpret_cast<const u32x4*>(src + c8 * 8), f);
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const float* wr = sw + (t * 3 + ci) * 64 + c8 * 8;
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[ci] = fmaf(f[k], wr[k], acc[ci]);
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const int64_t o = ((n * 3 + ci) * H + yy) * W + xx;
      const float d = pred[o] - target[o];
      dimg[o] = acc[ci] / kScale[ci] + l1_coef * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
  }
}

// ---- LPIPS head of one tap ------------------------------------------------------------------------------------------------
// f: [2B][HW][C] (images 0..B-1 = prediction, B..2B-1 = target); one group of C/8 lanes per pixel.
template <int G, bool BWD>
__global__ __launch_bounds__(256) void lpips_tap_kernel(const bf16_t* __restrict__ f, const float* __restrict__ lin,
                                                        float* __restrict__ val, bf16_t* __restrict__ gout, float coef_arg,
                                                        const float* __restrict__ gscale, int B, int HW) {
  const float coef = (BWD && gscale) ? coef_arg * gscale[0] : coef_arg;       // upstream scalar gradient stays on the device
  constexpr int C = G * 8;
  __shared__ float red[256 / 64];
  const int b = blockIdx.y;
  const int li = threadIdx.x % G, grp = threadIdx.x / G;
  constexpr int GPB = 256 / G;
  float w8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w8[k] = lin[li * 8 + k];
  float part = 0.f;
  for (int px = blockIdx.x * GPB + grp; px < HW; px += gridDim.x * GPB) {
    float a[8], t[8];
    unpack8f(*reinterpret_cast<const u32x4*>(f + ((size_t)b * HW + px) * C + li * 8), a);
    unpack8f(*reinterpret_cast<const u32x4*>(f + ((size_t)(B + b) * HW + px) * C + li * 8), t);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s0 = fmaf(a[k], a[k], s0); s1 = fmaf(t[k], t[k], s1); }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    const float r0 = sqrtf(s0), r1 = sqrtf(s1);
    const float i0 = 1.f / (r0 + 1e-10f), i1 = 1.f / (r1 + 1e-10f);
    float e8[8], d = 0.f, dotf = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      e8[k] = a[k] * i0 - t[k] * i1;
      d = fmaf(w8[k] * e8[k], e8[k], d);
      dotf = fmaf(2.f * w8[k] * e8[k], a[k], dotf);
    }
    if (!BWD) {
      part += d;
    } else {
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) dotf += __shfl_xor(dotf, o);
      // d(sum_k w_k e_k^2)/d a_c = 2 w_c e_c / n0 - a_c / (n0^2 r0) * sum_k 2 w_k e_k a_k   (n0 = r0 + eps; 0 at r0 = 0)
      const float k2 = r0 > 0.f ? dotf * i0 * i0 / r0 : 0.f;
      float g8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) g8[k] = a[k] > 0.f ? coef * (2.f * w8[k] * e8[k] * i0 - a[k] * k2) : 0.f;   // ReLU mask of the tap
      *reinterpret_cast<u32x4*>(gout + ((size_t)b * HW + px) * C + li * 8) = pack8f(g8);
    }
  }
  if (!BWD) {
    // every lane holds the contribution of its 8 channels over its group's pixels; reduce over the block
    float v = part;
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) val[(size_t)blockIdx.x * B + b] = red[0] + red[1] + red[2] + red[3];     // the block's slab (forward: val = workspace)
  }
}

__global__ __launch_bounds__(256) void l1_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, int64_t n, float inv_n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += fabsf(a[i] - b[i]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];      // out = workspace: one partial per block
}

unsigned grid_for(int64_t work) {
  int64_t g = (work + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  if (g < 1) g = 1;
  return (unsigned)g;
}
bool dims_ok(int N, int H, int W, int C) { return N > 0 && H > 0 && W > 0 && C > 0 && (C % 8) == 0; }

}  // namespace

extern "C" int orbit2_im2col3x3(const void* x, void* col, int N, int H, int W, int C, void* stream) {
  if (!x || !col || !dims_ok(N, H, W, C)) return O2_ERR_ARG;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for((int64_t)N * H * W * 9 * (C / 8))), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)col, N, H, W, C);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_col2im3x3(const void* dcol, const void* act, const void* tapg, void* out, int N, int H, int W,
                                int C, void* stream) {
  if (!dcol || !out || !dims_ok(N, H, W, C) || (tapg && !act)) return O2_ERR_ARG;
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for((int64_t)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dcol, (const bf16_t*)act, (const bf16_t*)tapg, (bf16_t*)out, N, H, W, C);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_maxpool2_fwd(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  if (!x || !y || !dims_ok(N, H, W, C) || (H & 1) || (W & 1)) return O2_ERR_ARG;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, N, H, W, C);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_maxpool2_bwd(const void* g, const void* x, const void* tapg, void* dz, int N, int H, int W, int C,
                                   void* stream) {
  if (!g || !x || !dz || !dims_ok(N, H, W, C) || (H & 1) || (W & 1)) return O2_ERR_ARG;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for((int64_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)g, (const bf16_t*)x, (const bf16_t*)tapg, (bf16_t*)dz, N, H, W, C);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_lpips_conv1_fwd(const float* img, const float* w1, const float* b1, void* out, int N, int H, int W,
                                      void* stream) {
  if (!img || !w1 || !b1 || !out || N <= 0 || H <= 0 || W <= 0) return O2_ERR_ARG;
  hipLaunchKernelGGL(lpips_conv1_fwd_kernel, dim3(grid_for((int64_t)N * H * W * 8)), dim3(256), 0, (hipStream_t)stream,
                     img, w1, b1, (bf16_t*)out, N, H, W);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_lpips_conv1_bwd(const void* dz, const float* w1, const float* pred, const float* target,
                                      float l1_coef, const float* gscale, float* dimg, int N, int H, int W, void* stream) {
  if (!dz || !w1 || !pred || !target || !dimg || N <= 0 || H <= 0 || W <= 0) return O2_ERR_ARG;
  hipLaunchKernelGGL(lpips_conv1_bwd_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dz, w1, pred, target, l1_coef, gscale, dimg, N, H, W);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

static int lpips_tap_blocks(int HW, int C) {
  int blocks = (HW * (C / 8) + 255) / 256;
  return blocks > 1024 ? 1024 : blocks;
}
template <bool BWD>
static int lpips_tap_launch(const void* f, const float* lin, float* val, void* gout, float coef, const float* gscale, int B,
                            int HW, int C, hipStream_t s) {
  const int blocks = lpips_tap_blocks(HW, C);
  dim3 grid((unsigned)blocks, (unsigned)B), block(256);
#define O2_TAP(G)                                                                                               \
  hipLaunchKernelGGL((lpips_tap_kernel<G, BWD>), grid, block, 0, s, (const bf16_t*)f, lin, val, (bf16_t*)gout, \
                     coef, gscale, B, HW)
  switch (C) {
    case 64: O2_TAP(8); break;
    case 128: O2_TAP(16); break;
    case 256: O2_TAP(32); break;
    case 512: O2_TAP(64); break;
    default: return O2_ERR_UNSUPPORTED;
  }
#undef O2_TAP
  O2_CHECK_LAUNCH();
  return O2_OK;
}

// forward sums: every block stores its partial in the workspace, the partials are added in a fixed order (round 4: these were
// fp32 atomics; the loss value differed in its last digits from run to run)
extern "C" int64_t orbit2_lpips_tap_ws_floats(int B, int HW, int C) {
  if (B <= 0 || HW <= 0 || C <= 0) return 0;
  return (int64_t)lpips_tap_blocks(HW, C) * B;
}
extern "C" int orbit2_lpips_tap_fwd(const void* feats, const float* lin, float* val, int B, int HW, int C, float* ws,
                                    void* stream) {
  if (!feats || !lin || !val || !ws || B <= 0 || HW <= 0) return O2_ERR_ARG;
  const int rc = lpips_tap_launch<false>(feats, lin, ws, nullptr, 0.f, nullptr, B, HW, C, (hipStream_t)stream);
  if (rc) return rc;
  o2_sum_parts(ws, lpips_tap_blocks(HW, C), B, val, B, 1.0f / (float)HW, 1, (hipStream_t)stream);      // val[b] += mean over pixels
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_lpips_tap_bwd(const void* feats, const float* lin, void* gout, float coef, const float* gscale, int B,
                                    int HW, int C, void* stream) {
  if (!feats || !lin || !gout || B <= 0 || HW <= 0) return O2_ERR_ARG;
  return lpips_tap_launch<true>(feats, lin, nullptr, gout, coef, gscale, B, HW, C, (hipStream_t)stream);
}

extern "C" int64_t orbit2_l1_mean_ws_floats(int64_t n) { return n > 0 ? (int64_t)grid_for(n) : 0; }
extern "C" int orbit2_l1_mean(const float* a, const float* b, float* out, int64_t n, float* ws, void* stream) {
  if (!a || !b || !out || !ws || n <= 0) return O2_ERR_ARG;
  const unsigned nb = grid_for(n);
  hipLaunchKernelGGL(l1_sum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, ws, n, 1.0f / (float)n);
  o2_sum_parts(ws, (int)nb, 1, out, 1, 1.0f / (float)n, 1, (hipStream_t)stream);                        // out += mean |a - b|
  O2_CHECK_LAUNCH();
  return O2_OK;
}
