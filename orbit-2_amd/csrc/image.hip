// Hi-res tail of Res_Slim_ViT (fp32, HBM-bound, tiny): unpatchify, 3x3 convs (+GELU+PixelShuffle),
// precipitation clamp, and the fused losses (mse / bayesian_tv with latitude and variable weights).
#include "common.h"
#include "../../include/orbit2_hip.h"

namespace {

inline int grid_for(int64_t n, int per_block) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  return (int)g;
}

// ---- unpatchify (res_slimvit.py:167-179): pure index permutation -------------------------------
// img[b][c][hh*p+pp][ww*p+qq] = t[b][ (((hh*wf + ww)*p + pp)*p + qq)*C + c ],  hf = h*s/p, wf = w*s/p
template <bool BWD>
__global__ __launch_bounds__(256) void unpatchify_kernel(bf16_t* __restrict__ t, float* __restrict__ img, int B, int C,
                                                         int Hh, int Wh, int p) {
  const int64_t per = (int64_t)C * Hh * Wh;
  const int64_t n = (int64_t)B * per;
  const int wf = Wh / p;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / per);
    int64_t r = e - (int64_t)b * per;
    if (!BWD) {
      // e enumerates img elements
      const int c = (int)(r / ((int64_t)Hh * Wh));
      r -= (int64_t)c * Hh * Wh;
      const int y = (int)(r / Wh), x = (int)(r - (int64_t)y * Wh);
      const int hh = y / p, pp = y - hh * p, ww = x / p, qq = x - ww * p;
      const int64_t ti = ((((int64_t)hh * wf + ww) * p + pp) * p + qq) * C + c;
      img[e] = bf2f(t[(int64_t)b * per + ti]);
    } else {
      // e enumerates t elements
      const int c = (int)(r % C);
      int64_t q = r / C;
      const int qq = (int)(q % p); q /= p;
      const int pp = (int)(q % p); q /= p;
      const int ww = (int)(q % wf);
      const int hh = (int)(q / wf);
      t[e] = f2bf(img[(int64_t)b * per + ((int64_t)c * Hh + (hh * p + pp)) * Wh + (ww * p + qq)]);
    }
  }
}

// ---- 3x3 conv forward: one thread per output pixel, COB output channels per pass ---------------
constexpr int CONV_MAXCIN = 8;
constexpr int COB = 16;
__global__ __launch_bounds__(256) void conv3x3_fwd_kernel(const float* __restrict__ in, const int* __restrict__ cidx,
                                                          int in_ctotal, const float* __restrict__ wgt,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          float* __restrict__ pre, const float* __restrict__ addend,
                                                          int Ha, int Wa, int B, int Cin, int Cout, int H, int W,
                                                          int mode, int r) {
  __shared__ float sw[COB * CONV_MAXCIN * 9 + COB];
  const int co0 = blockIdx.y * COB;
  const int nco = (Cout - co0) < COB ? (Cout - co0) : COB;
  for (int e = threadIdx.x; e < nco * Cin * 9; e += 256) sw[e] = wgt[(size_t)co0 * Cin * 9 + e];
  for (int e = threadIdx.x; e < nco; e += 256) sw[COB * CONV_MAXCIN * 9 + e] = bias[co0 + e];
  __syncthreads();
  const int64_t npix = (int64_t)B * H * W;
  for (int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x; px < npix; px += (int64_t)gridDim.x * 256) {
    const int b = (int)(px / ((int64_t)H * W));
    const int rem = (int)(px - (int64_t)b * H * W);
    const int y = rem / W, x = rem - y * W;
    float nb[CONV_MAXCIN][9];
#pragma unroll
    for (int ci = 0; ci < CONV_MAXCIN; ++ci) {
      if (ci < Cin) {
        const int cs = cidx ? cidx[ci] : ci;
        const float* pl = in + ((size_t)b * in_ctotal + cs) * H * W;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
          nb[ci][k] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? pl[(size_t)yy * W + xx] : 0.f;
        }
      }
    }
    for (int co = 0; co < nco; ++co) {
      float acc = sw[COB * CONV_MAXCIN * 9 + co];
#pragma unroll
      for (int ci = 0; ci < CONV_MAXCIN; ++ci) {
        if (ci < Cin) {
#pragma unroll
          for (int k = 0; k < 9; ++k) acc = fmaf(sw[(co * Cin + ci) * 9 + k], nb[ci][k], acc);
        }
      }
      const int cg = co0 + co;
      if (mode == 0) {
        if (addend) acc += addend[(((size_t)b * Cout + cg) * Ha + y) * Wa + x];
        out[(((size_t)b * Cout + cg) * H + y) * W + x] = acc;
      } else {
        pre[(((size_t)b * Cout + cg) * H + y) * W + x] = acc;
        const int rr = r * r;
        const int oc = cg / rr, sub = cg - oc * rr;
        out[(((size_t)b * (Cout / rr) + oc) * (H * r) + (y * r + sub / r)) * (size_t)(W * r) + (x * r + sub % r)] =
            gelu_f(acc);
      }
    }
  }
}

// dpre value at (b, co, y, x) for either mode
__device__ __forceinline__ float conv_dpre(const float* __restrict__ dout, const float* __restrict__ pre, int b,
                                           int co, int y, int x, int Cout, int H, int W, int mode, int r) {
  if (mode == 0) return dout[(((size_t)b * Cout + co) * H + y) * W + x];
  const int rr = r * r;
  const int oc = co / rr, sub = co - oc * rr;
  const float d =
      dout[(((size_t)b * (Cout / rr) + oc) * (H * r) + (y * r + sub / r)) * (size_t)(W * r) + (x * r + sub % r)];
  return d * dgelu_f(pre[(((size_t)b * Cout + co) * H + y) * W + x]);
}

// ---- input gradient: one thread per (b, ci, y, x) ------------------------------------------------
__global__ __launch_bounds__(256) void conv3x3_bwd_data_kernel(const float* __restrict__ dout,
                                                               const float* __restrict__ pre,
                                                               const float* __restrict__ wgt, float* __restrict__ din,
                                                               int B, int Cin, int Cout, int H, int W, int mode,
                                                               int r) {
  const int64_t n = (int64_t)B * Cin * H * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    int64_t q = e;
    const int x = (int)(q % W); q /= W;
    const int y = (int)(q % H); q /= H;
    const int ci = (int)(q % Cin);
    const int b = (int)(q / Cin);
    float acc = 0.f;
    for (int co = 0; co < Cout; ++co) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int yy = y + 1 - k / 3, xx = x + 1 - k % 3;  // output pixel that saw (y,x) through tap k
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
          acc = fmaf(conv_dpre(dout, pre, b, co, yy, xx, Cout, H, W, mode, r), wgt[((size_t)co * Cin + ci) * 9 + k],
                     acc);
      }
    }
    din[e] = acc;
  }
}

// ---- weight / bias gradient: 16x16 pixel tile per block, one thread per (co, ci, tap) entry -------
// Every block stores its Cout*Cin*9 + Cout partial sums in its own slab of `ws`; o2_sum_parts adds the slabs in a fixed
// order (round 4: the fp32 atomics this replaced made the step's last digits differ from run to run).
constexpr int TW = 16, TH = 16;
__global__ __launch_bounds__(256) void conv3x3_bwd_weight_kernel(const float* __restrict__ dout,
                                                                 const float* __restrict__ pre,
                                                                 const float* __restrict__ in,
                                                                 const int* __restrict__ cidx, int in_ctotal,
                                                                 float* __restrict__ ws,
                                                                 int B, int Cin, int Cout, int H, int W, int mode,
                                                                 int r) {
  float* dw = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ((size_t)Cout * Cin * 9 + Cout);
  float* dbias = dw + (size_t)Cout * Cin * 9;
  __shared__ float sd[COB][TH][TW];
  __shared__ float si[CONV_MAXCIN][TH + 2][TW + 2];
  const int tiles_x = (W + TW - 1) / TW;
  const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
  const int b = blockIdx.y;
  for (int e = threadIdx.x; e < Cin * (TH + 2) * (TW + 2); e += 256) {
    const int ci = e / ((TH + 2) * (TW + 2));
    const int rem = e - ci * (TH + 2) * (TW + 2);
    const int yy = ty0 + rem / (TW + 2) - 1, xx = tx0 + rem % (TW + 2) - 1;
    const int cs = cidx ? cidx[ci] : ci;
    si[ci][rem / (TW + 2)][rem % (TW + 2)] =
        (yy >= 0 && yy < H && xx >= 0 && xx < W) ? in[(((size_t)b * in_ctotal + cs) * H + yy) * W + xx] : 0.f;
  }
  for (int co0 = 0; co0 < Cout; co0 += COB) {
    const int nco = (Cout - co0) < COB ? (Cout - co0) : COB;
    __syncthreads();
    for (int e = threadIdx.x; e < nco * TH * TW; e += 256) {
      const int co = e / (TH * TW);
      const int rem = e - co * TH * TW;
      const int y = ty0 + rem / TW, x = tx0 + rem % TW;
      sd[co][rem / TW][rem % TW] =
          (y < H && x < W) ? conv_dpre(dout, pre, b, co0 + co, y, x, Cout, H, W, mode, r) : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nco * Cin * 9; e += 256) {
      const int co = e / (Cin * 9);
      const int rem = e - co * Cin * 9;
      const int ci = rem / 9, k = rem - ci * 9;
      const int ky = k / 3, kx = k - ky * 3;
      float s = 0.f;
      for (int y = 0; y < TH; ++y)
#pragma unroll
        for (int x = 0; x < TW; ++x) s = fmaf(sd[co][y][x], si[ci][y + ky][x + kx], s);
      dw[((size_t)(co0 + co) * Cin + ci) * 9 + k] = s;
    }
    for (int co = threadIdx.x; co < nco; co += 256) {
      float s = 0.f;
      for (int y = 0; y < TH; ++y)
        for (int x = 0; x < TW; ++x) s += sd[co][y][x];
      dbias[co0 + co] = s;
    }
  }
}

__global__ __launch_bounds__(256) void clamp_channel_kernel(float* __restrict__ img, const float* __restrict__ ref,
                                                            float* __restrict__ dimg, int B, int C, int HW,
                                                            int chan) {
  const int64_t n = (int64_t)B * HW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / HW);
    const size_t off = ((size_t)b * C + chan) * HW + (e - (int64_t)b * HW);
    if (dimg) { if (!(ref[off] > 0.f)) dimg[off] = 0.f; }
    else img[off] = fmaxf(img[off], 0.f);
  }
}

// ---- losses ---------------------------------------------------------------------------------------
// err(b,c,i,j) = [(p-t)^2 + 0.02*(dv + dh + 0.7*d1 + 0.7*d2)] * latw[i] * chanw[c]     (kind 1; kind 0: no TV)
//   dv = |p[i+1][j]-p[i][j]| (i<H-1)        dh = |p[i][j+1]-p[i][j]| (j<W-1)
//   d1 = |p[i+1][j+1]-p[i][j]| (i<H-1,j<W-1)   d2 = |p[i+1][j-1]-p[i][j]| (i<H-1, j>=1)
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       int Ht, int Wt, const float* __restrict__ latw,
                                                       const float* __restrict__ chanw, float* __restrict__ part,
                                                       float* __restrict__ part2, int C, int H, int W, int kind) {
  const int plane = blockIdx.y;  // b*C + c
  const int c = plane % C, b = plane / C;
  const float* p = pred + (size_t)plane * H * W;
  const float* t = tgt + ((size_t)b * C + c) * Ht * Wt;
  const float cw = chanw ? chanw[c] : 1.f;
  float s = 0.f, s2 = 0.f;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < H * W; e += gridDim.x * 256) {
    const int i = e / W, j = e - i * W;
    const float pv = p[e];
    const float tv0 = t[(size_t)i * Wt + j];
    const float d = pv - tv0;
    float err = d * d;
    if (kind == 2) {   // image-gradient term (metrics/functional.py:59-114): forward differences, last row/col zero
      if (j < W - 1) s2 += fabsf((t[(size_t)i * Wt + j + 1] - tv0) - (p[e + 1] - pv));
      if (i < H - 1) s2 += fabsf((t[(size_t)(i + 1) * Wt + j] - tv0) - (p[e + W] - pv));
    }
    if (kind == 1) {
      float tv = 0.f;
      if (i < H - 1) {
        tv += fabsf(p[e + W] - pv);
        if (j < W - 1) tv += 0.7f * fabsf(p[e + W + 1] - pv);
        if (j >= 1) tv += 0.7f * fabsf(p[e + W - 1] - pv);
      }
      if (j < W - 1) tv += fabsf(p[e + 1] - pv);
      err += 0.02f * tv;
    }
    s += err * (latw ? latw[i] : 1.f) * cw;
  }
  s = wave_sum(s);
  s2 = wave_sum(s2);
  __shared__ float sw[8];
  if ((threadIdx.x & 63) == 0) { sw[threadIdx.x >> 6] = s; sw[4 + (threadIdx.x >> 6)] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[(size_t)plane * gridDim.x + blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
    part2[(size_t)plane * gridDim.x + blockIdx.x] = sw[4] + sw[5] + sw[6] + sw[7];
  }
}

__global__ void loss_final_kernel(const float* __restrict__ part, const float* __restrict__ part2, int nblk, int B,
                                  int C, int HW, const float* __restrict__ chanw, int kind, float* __restrict__ out) {
  // one wave; out[c] = mean over (b, pixels) of channel c; out[C] = mean over everything
  float tot = 0.f;
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (int e = threadIdx.x; e < B * nblk; e += 64) {
      const int b = e / nblk, k = e - b * nblk;
      s += part[((size_t)(b * C + c)) * nblk + k];
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) out[c] = s / ((float)B * (float)HW);
    tot += s;
  }
  float agg = tot / ((float)B * (float)C * (float)HW);
  if (kind == 2) {   // + 0.1 * mean(|grad diff|) * mean(channel weights)
    float s2 = 0.f;
    for (int e = threadIdx.x; e < B * C * nblk; e += 64) s2 += part2[e];
    s2 = wave_sum(s2);
    float cwm = 1.f;
    if (chanw) { cwm = 0.f; for (int c = 0; c < C; ++c) cwm += chanw[c]; cwm /= (float)C; }
    agg += 0.1f * (s2 / ((float)B * (float)C * (float)HW)) * cwm;
  }
  if (threadIdx.x == 0) out[C] = agg;
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       int Ht, int Wt, const float* __restrict__ latw,
                                                       const float* __restrict__ chanw,
                                                       const float* __restrict__ gscale, float* __restrict__ dpred,
                                                       int B, int C, int H, int W, int kind) {
  const int64_t n = (int64_t)B * C * H * W;
  const float g0 = gscale[0] / (float)n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    int64_t q = e;
    const int j = (int)(q % W); q /= W;
    const int i = (int)(q % H); q /= H;
    const int c = (int)(q % C);
    const int b = (int)(q / C);
    const float* p = pred + ((size_t)b * C + c) * H * W;
    const int o = i * W + j;
    const float pv = p[o];
    const float wi = latw ? latw[i] : 1.f;
    const float wim = (latw && i > 0) ? latw[i - 1] : 1.f;
    float g = 2.f * (pv - tgt[(((size_t)b * C + c) * Ht + i) * Wt + j]) * wi;
    if (kind == 2) {
      const float* t = tgt + ((size_t)b * C + c) * Ht * Wt;
      const float tv0 = t[(size_t)i * Wt + j];
      float gs = 0.f;
      if (j < W - 1) gs += sgn((t[(size_t)i * Wt + j + 1] - tv0) - (p[o + 1] - pv));
      if (j > 0) gs -= sgn((tv0 - t[(size_t)i * Wt + j - 1]) - (pv - p[o - 1]));
      if (i < H - 1) gs += sgn((t[(size_t)(i + 1) * Wt + j] - tv0) - (p[o + W] - pv));
      if (i > 0) gs -= sgn((tv0 - t[(size_t)(i - 1) * Wt + j]) - (pv - p[o - W]));
      float cwm = 1.f;
      if (chanw) { cwm = 0.f; for (int cc = 0; cc < C; ++cc) cwm += chanw[cc]; cwm /= (float)C; }
      dpred[e] = (g * (chanw ? chanw[c] : 1.f) + 0.1f * cwm * gs) * g0;
      continue;
    }
    if (kind == 1) {
      float tv = 0.f;
      // terms stored at row i (weight wi) in which p[i][j] is the subtrahend
      if (i < H - 1) {
        tv -= sgn(p[o + W] - pv) * wi;
        if (j < W - 1) tv -= 0.7f * sgn(p[o + W + 1] - pv) * wi;
        if (j >= 1) tv -= 0.7f * sgn(p[o + W - 1] - pv) * wi;
      }
      if (j < W - 1) tv -= sgn(p[o + 1] - pv) * wi;
      // terms in which p[i][j] is the minuend
      if (j > 0) tv += sgn(pv - p[o - 1]) * wi;                              // dh stored at (i, j-1)
      if (i > 0) {
        tv += sgn(pv - p[o - W]) * wim;                                      // dv stored at (i-1, j)
        if (j > 0) tv += 0.7f * sgn(pv - p[o - W - 1]) * wim;                // d1 stored at (i-1, j-1)
        if (j < W - 1) tv += 0.7f * sgn(pv - p[o - W + 1]) * wim;            // d2 stored at (i-1, j+1)
      }
      g += 0.02f * tv;
    }
    dpred[e] = g * (chanw ? chanw[c] : 1.f) * g0;
  }
}

}  // namespace

extern "C" int orbit2_unpatchify_fwd(const void* t, float* img, int B, int C, int h, int w, int p, int s,
                                     void* stream) {
  if (!t || !img || B <= 0 || C <= 0 || p <= 0 || s <= 0 || (h * s) % p || (w * s) % p) return O2_ERR_ARG;
  const int64_t n = (int64_t)B * C * h * s * w * s;
  hipLaunchKernelGGL(unpatchify_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)t,
                     img, B, C, h * s, w * s, p);
  O2_CHECK_LAUNCH();
  return O2_OK;
}
extern "C" int orbit2_unpatchify_bwd(const float* dimg, void* dt, int B, int C, int h, int w, int p, int s,
                                     void* stream) {
  if (!dimg || !dt || B <= 0 || C <= 0 || p <= 0 || s <= 0 || (h * s) % p || (w * s) % p) return O2_ERR_ARG;
  const int64_t n = (int64_t)B * C * h * s * w * s;
  hipLaunchKernelGGL(unpatchify_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dt,
                     (float*)dimg, B, C, h * s, w * s, p);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_conv3x3_fwd(const float* in, const int* chan_idx, int in_ctotal, const float* weight,
                                  const float* bias, float* out, float* pre, const float* addend, int Ha, int Wa,
                                  int B, int Cin, int Cout, int H, int W, int mode, int r, void* stream) {
  if (!in || !weight || !bias || !out || B <= 0 || Cin <= 0 || Cin > CONV_MAXCIN || Cout <= 0 || H <= 0 || W <= 0)
    return O2_ERR_ARG;
  if (mode == 1 && (!pre || r <= 0 || Cout % (r * r))) return O2_ERR_ARG;
  if (mode != 0 && mode != 1) return O2_ERR_ARG;
  if (addend && (mode != 0 || Ha < H || Wa < W)) return O2_ERR_ARG;
  dim3 grid(grid_for((int64_t)B * H * W, 256), (Cout + COB - 1) / COB);
  hipLaunchKernelGGL(conv3x3_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, chan_idx, in_ctotal, weight,
                     bias, out, pre, addend, Ha, Wa, B, Cin, Cout, H, W, mode, r);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int64_t orbit2_conv3x3_bwd_ws_floats(int B, int Cin, int Cout, int H, int W) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  return (int64_t)B * ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * ((int64_t)Cout * Cin * 9 + Cout);
}

extern "C" int orbit2_conv3x3_bwd(const float* dout, const float* in, const int* chan_idx, int in_ctotal,
                                  const float* weight, const float* pre, float* din, float* dweight, float* dbias,
                                  int B, int Cin, int Cout, int H, int W, int mode, int r, float* ws, void* stream) {
  if (!dout || !in || !weight || !dweight || !dbias || !ws || B <= 0 || Cin <= 0 || Cin > CONV_MAXCIN || Cout <= 0)
    return O2_ERR_ARG;
  if (mode == 1 && (!pre || r <= 0 || Cout % (r * r))) return O2_ERR_ARG;
  if (mode != 0 && mode != 1) return O2_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (din) {
    hipLaunchKernelGGL(conv3x3_bwd_data_kernel, dim3(grid_for((int64_t)B * Cin * H * W, 256)), dim3(256), 0, s, dout,
                       pre, weight, din, B, Cin, Cout, H, W, mode, r);
    O2_CHECK_LAUNCH();
  }
  dim3 grid(((W + TW - 1) / TW) * ((H + TH - 1) / TH), B);
  hipLaunchKernelGGL(conv3x3_bwd_weight_kernel, grid, dim3(256), 0, s, dout, pre, in, chan_idx, in_ctotal, ws,
                     B, Cin, Cout, H, W, mode, r);
  const int nparts = (int)(grid.x * grid.y);
  const int64_t nw = (int64_t)Cout * Cin * 9, slab = nw + Cout;
  o2_sum_parts(ws, nparts, slab, dweight, nw, 1.0f, 1, s);          // dweight / dbias += (as the atomics did)
  o2_sum_parts(ws + nw, nparts, slab, dbias, Cout, 1.0f, 1, s);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_clamp_channel(float* img, int B, int C, int HW, int chan, void* stream) {
  if (!img || B <= 0 || C <= 0 || HW <= 0 || chan < 0 || chan >= C) return O2_ERR_ARG;
  hipLaunchKernelGGL(clamp_channel_kernel, dim3(grid_for((int64_t)B * HW, 256)), dim3(256), 0, (hipStream_t)stream,
                     img, (const float*)nullptr, (float*)nullptr, B, C, HW, chan);
  O2_CHECK_LAUNCH();
  return O2_OK;
}
extern "C" int orbit2_clamp_channel_bwd(const float* img_clamped, float* dimg, int B, int C, int HW, int chan,
                                        void* stream) {
  if (!img_clamped || !dimg || B <= 0 || C <= 0 || HW <= 0 || chan < 0 || chan >= C) return O2_ERR_ARG;
  hipLaunchKernelGGL(clamp_channel_kernel, dim3(grid_for((int64_t)B * HW, 256)), dim3(256), 0, (hipStream_t)stream,
                     (float*)nullptr, img_clamped, dimg, B, C, HW, chan);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

static const int LOSS_NBLK = 64;
extern "C" int orbit2_loss_fwd(const float* pred, const float* target, int Ht, int Wt, const float* lat_w,
                               const float* chan_w, float* out, float* ws, int B, int C, int H, int W, int kind,
                               void* stream) {
  if (!pred || !target || !out || !ws || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ht < H || Wt < W) return O2_ERR_ARG;
  if (kind < 0 || kind > 2) return O2_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* ws2 = ws + (size_t)B * C * LOSS_NBLK;
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(LOSS_NBLK, B * C), dim3(256), 0, s, pred, target, Ht, Wt, lat_w, chan_w, ws,
                     ws2, C, H, W, kind);
  O2_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, ws, ws2, LOSS_NBLK, B, C, H * W, chan_w, kind, out);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_loss_bwd(const float* pred, const float* target, int Ht, int Wt, const float* lat_w,
                               const float* chan_w, const float* gscale, float* dpred, int B, int C, int H, int W,
                               int kind, void* stream) {
  if (!pred || !target || !gscale || !dpred || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ht < H || Wt < W)
    return O2_ERR_ARG;
  if (kind < 0 || kind > 2) return O2_ERR_ARG;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid_for((int64_t)B * C * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     pred, target, Ht, Wt, lat_w, chan_w, gscale, dpred, B, C, H, W, kind);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

// ---- evaluation metrics (metrics/functional.py:219-324: mae, rmse, acc, pearson, mean_bias) ------------------------------
// twelve sums per (b, c) image over a = pred - clim, b = target - clim (clim optional, [C][H][W]), w = lat_w[y] or 1:
//   0 sum a, 1 sum b, 2 sum a^2, 3 sum b^2, 4 sum a*b, 5 sum w (a-b)^2, 6 sum w |a-b|,
//   7 sum w a, 8 sum w b, 9 sum w a*b, 10 sum w a^2, 11 sum w b^2
// per-thread fp32 partials, double accumulation across the grid.
namespace {
constexpr int EVAL_NM = 12;
__global__ __launch_bounds__(256) void eval_moments_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           int Ht, int Wt, const float* __restrict__ lat_w,
                                                           const float* __restrict__ clim, double* __restrict__ out,
                                                           int C, int H, int W) {
  __shared__ float red[4][EVAL_NM];
  const int bc = blockIdx.y;
  const float* p = pred + (size_t)bc * H * W;
  const float* t = target + (size_t)bc * Ht * Wt;
  const float* cl = clim ? clim + (size_t)(bc % C) * H * W : nullptr;
  float s[EVAL_NM];
#pragma unroll
  for (int k = 0; k < EVAL_NM; ++k) s[k] = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W;
    const float c0 = cl ? cl[i] : 0.f;
    const float a = p[i] - c0, b = t[(size_t)y * Wt + x] - c0;
    const float w = lat_w ? lat_w[y] : 1.f;
    const float d = a - b;
    s[0] += a; s[1] += b; s[2] += a * a; s[3] += b * b; s[4] += a * b;
    s[5] += w * d * d; s[6] += w * fabsf(d);
    s[7] += w * a; s[8] += w * b; s[9] += w * a * b; s[10] += w * a * a; s[11] += w * b * b;
  }
#pragma unroll
  for (int k = 0; k < EVAL_NM; ++k) {
    const float v = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < EVAL_NM) {
    const int k = threadIdx.x;
    atomicAdd(out + (size_t)bc * EVAL_NM + k,
              (double)red[0][k] + (double)red[1][k] + (double)red[2][k] + (double)red[3][k]);
  }
}
}  // namespace

extern "C" int orbit2_eval_moments(const float* pred, const float* target, int Ht, int Wt, const float* lat_w,
                                   const float* clim, double* out, int B, int C, int H, int W, void* stream) {
  if (!pred || !target || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ht < H || Wt < W) return O2_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(double) * (size_t)B * C * EVAL_NM, s) != hipSuccess) return O2_ERR_LAUNCH;
  int nblk = (H * W + 256 * 8 - 1) / (256 * 8);
  if (nblk > 64) nblk = 64;
  hipLaunchKernelGGL(eval_moments_kernel, dim3(nblk, B * C), dim3(256), 0, s, pred, target, Ht, Wt, lat_w, clim, out, C,
                     H, W);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

// ---- position-embedding table of a step: bicubic re-grid + resolution embedding --------------------------------------------
// out[(oy nw + ox)][:] = bicubic(pe)[oy][ox][:] + sw[:] * res + sb[:]
// (components/pos_embed.py:103-138 interpolate_pos_embed_on_the_fly: the [1, L0, D] table seen as a channel-last [oh][ow][D] grid,
// F.interpolate(mode="bicubic", align_corners=False) to [nh][nw], only when the heights differ; res_slimvit.py:277-281: + the
// Linear(1, D) spatial embedding of the scalar resolution.)  fp32 throughout, the table is a parameter.
// Bicubic as ATen defines it: source coordinate s = (in / out) (o + 0.5) - 0.5 (not clamped), base = floor(s), t = s - base, taps
// base - 1 .. base + 2 with indices clamped to the grid, cubic-convolution weights with A = -0.75; a pixel = the four row
// interpolants (left to right) combined top to bottom.  Channel-last, so a pixel's D values are contiguous: one workgroup per
// pixel, lanes over D in float4.
namespace {
constexpr int PE_MAX_SIDE = 2048;

__device__ __forceinline__ void cubic_taps(int o, int n_in, int n_out, int idx[4], float w[4]) {
  const float scale = (float)n_in / (float)n_out;
  const float s = scale * ((float)o + 0.5f) - 0.5f;
  const float fl = floorf(s);
  const float t = s - fl;
  const int base = (int)fl;
  constexpr float A = -0.75f;
  const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int i = base - 1 + k;
    idx[k] = i < 0 ? 0 : (i > n_in - 1 ? n_in - 1 : i);
  }
}

template <bool SAME>
__global__ __launch_bounds__(256) void posembed_fwd_kernel(const float* __restrict__ pe, const float* __restrict__ sw,
                                                           const float* __restrict__ sb, float res, float* __restrict__ out,
                                                           int oh, int ow, int nh, int nw, int D) {
  const int pix = blockIdx.x, oy = pix / nw, ox = pix - oy * nw;
  int iy[4], ix[4];
  float wy[4], wx[4];
  if (!SAME) {
    cubic_taps(oy, oh, nh, iy, wy);
    cubic_taps(ox, ow, nw, ix, wx);
  }
  float* o = out + (size_t)pix * D;
  for (int d = threadIdx.x * 4; d < D; d += 256 * 4) {
    float4 acc;
    if (SAME) {
      acc = *reinterpret_cast<const float4*>(pe + (size_t)pix * D + d);
    } else {
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float* row = pe + ((size_t)iy[a] * ow) * D + d;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float4 v = *reinterpret_cast<const float4*>(row + (size_t)ix[b] * D);
          r.x += v.x * wx[b]; r.y += v.y * wx[b]; r.z += v.z * wx[b]; r.w += v.w * wx[b];
        }
        acc.x += r.x * wy[a]; acc.y += r.y * wy[a]; acc.z += r.z * wy[a]; acc.w += r.w * wy[a];
      }
    }
    if (sw) {
      const float4 w4 = *reinterpret_cast<const float4*>(sw + d), b4 = *reinterpret_cast<const float4*>(sb + d);
      acc.x += w4.x * res + b4.x; acc.y += w4.y * res + b4.y; acc.z += w4.z * res + b4.z; acc.w += w4.w * res + b4.w;
    }
    *reinterpret_cast<float4*>(o + d) = acc;
  }
}

// transpose of the re-grid: dpe[iy][ix][:] = sum_{oy, ox} Wy[oy][iy] Wx[ox][ix] dout[oy][ox][:].  One workgroup per SOURCE pixel:
// it first tabulates, in LDS, the weight every output row / column gives this source row / column (dense over nh / nw, mostly
// zero: a source row feeds ~4 in/out ... 4 out/in output rows, more at a clamped border), then walks the non-zero pairs in
// ascending (oy, ox) order -- a fixed summation order, no atomics (the same bits on every run and every rank).
__global__ __launch_bounds__(256) void posembed_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dpe, int oh, int ow,
                                                           int nh, int nw, int D) {
  __shared__ float wyd[PE_MAX_SIDE], wxd[PE_MAX_SIDE];
  const int pix = blockIdx.x, sy = pix / ow, sx = pix - sy * ow;
  for (int o = threadIdx.x; o < nh; o += 256) {
    int idx[4];
    float w[4], s = 0.f;
    cubic_taps(o, oh, nh, idx, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (idx[k] == sy) ? w[k] : 0.f;
    wyd[o] = s;
  }
  for (int o = threadIdx.x; o < nw; o += 256) {
    int idx[4];
    float w[4], s = 0.f;
    cubic_taps(o, ow, nw, idx, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (idx[k] == sx) ? w[k] : 0.f;
    wxd[o] = s;
  }
  __syncthreads();
  for (int d = threadIdx.x * 4; d < D; d += 256 * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = 0; oy < nh; ++oy) {
      const float wy = wyd[oy];
      if (wy == 0.f) continue;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int ox = 0; ox < nw; ++ox) {
        const float wx = wxd[ox];
        if (wx == 0.f) continue;
        const float4 v = *reinterpret_cast<const float4*>(dout + ((size_t)oy * nw + ox) * D + d);
        r.x += v.x * wx; r.y += v.y * wx; r.z += v.z * wx; r.w += v.w * wx;
      }
      acc.x += r.x * wy; acc.y += r.y * wy; acc.z += r.z * wy; acc.w += r.w * wy;
    }
    *reinterpret_cast<float4*>(dpe + (size_t)pix * D + d) = acc;
  }
}
}  // namespace

extern "C" int orbit2_posembed_fwd(const float* pe, const float* sw, const float* sb, float res, float* out, int oh, int ow,
                                   int nh, int nw, int D, void* stream) {
  if (!pe || !out || (sw == nullptr) != (sb == nullptr) || oh <= 0 || ow <= 0 || nh <= 0 || nw <= 0 || D <= 0 || (D & 3))
    return O2_ERR_ARG;
  if (nh > PE_MAX_SIDE || nw > PE_MAX_SIDE || oh > PE_MAX_SIDE || ow > PE_MAX_SIDE) return O2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (oh == nh) {                 // the reference re-grids only when the heights differ (pos_embed.py:117)
    if (ow != nw) return O2_ERR_ARG;
    hipLaunchKernelGGL((posembed_fwd_kernel<true>), dim3(nh * nw), dim3(256), 0, s, pe, sw, sb, res, out, oh, ow, nh, nw, D);
  } else {
    hipLaunchKernelGGL((posembed_fwd_kernel<false>), dim3(nh * nw), dim3(256), 0, s, pe, sw, sb, res, out, oh, ow, nh, nw, D);
  }
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_posembed_bwd(const float* dout, float* dpe, int oh, int ow, int nh, int nw, int D, void* stream) {
  if (!dout || !dpe || oh <= 0 || ow <= 0 || nh <= 0 || nw <= 0 || D <= 0 || (D & 3)) return O2_ERR_ARG;
  if (nh > PE_MAX_SIDE || nw > PE_MAX_SIDE || oh > PE_MAX_SIDE || ow > PE_MAX_SIDE) return O2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (oh == nh) {
    if (ow != nw) return O2_ERR_ARG;
    if (hipMemcpyAsync(dpe, dout, sizeof(float) * (size_t)nh * nw * D, hipMemcpyDeviceToDevice, s) != hipSuccess) return O2_ERR_LAUNCH;
    return O2_OK;
  }
  hipLaunchKernelGGL(posembed_bwd_kernel, dim3(oh * ow), dim3(256), 0, s, dout, dpe, oh, ow, nh, nw, D);
  O2_CHECK_LAUNCH();
  return O2_OK;
}
