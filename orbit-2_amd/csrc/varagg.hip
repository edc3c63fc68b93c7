// Folded per-variable patch-embedding + variable-aggregation cross-attention (gfx950).
//
// Reference math (res_slimvit.py:250-265 + :205-230, patch_embed.py:44-52, attention.py:132-176):
//   e_v  = W_v p_v + b_v + var_embed_v            (p_v = the token's 2x2 patch of variable v, 4 values)
//   k_v  = Wk e_v, val_v = Wv e_v, q = Wq var_query           (no bias, no dropout)
//   a_hv = softmax_v( scale * q_h . k_vh ),   z_h = sum_v a_hv val_vh ,   out = proj(z)
// Because e_v is affine in the 4 patch values, with pt_v = (p_v, 1) in R^5:
//   score_hv = sum_c stab[h][v][c] * pt_v[c]          stab = scale * (Wk_h^T q_h) . [W_v | b_v + var_embed_v]
//   z[i]     = sum_v a_{h(i)v} sum_c gtab[v][c][i] * pt_v[c]      gtab[v][c][:] = Wv [W_v | b_v+var_embed_v][:,c]
// The two tables depend on the weights only (built per step by small fp32 GEMMs with autograd); this kernel
// does the per-token work and never materialises the [B,V,L,D] tensor (1.16 GB/sample at interm_1b) nor runs
// the M = B*L*V "kv" GEMM.  Exact algebra; only the rounding order differs from the reference.
#include "common.h"
#include "../../include/orbit2_hip.h"

namespace {

constexpr int VA_T = 16;      // tokens per workgroup
constexpr int VA_MAXV = 32;   // max variables
constexpr int VA_MAXH = 32;   // max heads

// LDS: pt[T][V][5], aw[T][H][V]
template <bool BWD>
__global__ __launch_bounds__(256) void varagg_kernel(const float* __restrict__ x, const float* __restrict__ stab,
                                                     const float* __restrict__ gtab, bf16_t* __restrict__ z,
                                                     float* __restrict__ attw, const bf16_t* __restrict__ dz,
                                                     float* __restrict__ dstab, float* __restrict__ dgtab, int B,
                                                     int V, int h, int w, int H, int D) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* pt = sm;                       // [T][V][5]
  float* aw = pt + VA_T * V * 5;        // [T][H][V]
  float* da = aw + VA_T * H * V;        // [T][H][V] (backward only)
  const int tid = threadIdx.x;
  const int Lw = w / 2, L = (h / 2) * Lw;
  const int64_t tok0 = (int64_t)blockIdx.x * VA_T;
  const int64_t ntok = (int64_t)B * L;
  const int dh = D / H;

  // ---- patches ---------------------------------------------------------------------------------
  for (int e = tid; e < VA_T * V * 5; e += 256) {
    const int c = e % 5, v = (e / 5) % V, t = e / (5 * V);
    const int64_t tok = tok0 + t;
    float val = 0.f;
    if (tok < ntok) {
      if (c == 4) val = 1.f;
      else {
        const int b = (int)(tok / L), l = (int)(tok - (int64_t)b * L);
        const int pr = l / Lw, pc = l - pr * Lw;
        val = x[(((size_t)b * V + v) * h + (2 * pr + (c >> 1))) * w + 2 * pc + (c & 1)];
      }
    }
    pt[e] = val;
  }
  __syncthreads();
  // ---- scores + softmax over variables (one thread per (token, head)) -----------------------------
  if (!BWD) {
    for (int e = tid; e < VA_T * H; e += 256) {
      const int hh = e % H, t = e / H;
      const float* st = stab + (size_t)hh * V * 5;
      const float* pp = pt + (size_t)t * V * 5;
      float mx = -1e30f;
      for (int v = 0; v < V; ++v) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) s += st[v * 5 + c] * pp[v * 5 + c];
        mx = fmaxf(mx, s);
      }
      float sum = 0.f;
      for (int v = 0; v < V; ++v) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) s += st[v * 5 + c] * pp[v * 5 + c];
        sum += __expf(s - mx);
      }
      const float inv = 1.f / sum;
      const int64_t tok = tok0 + t;
      for (int v = 0; v < V; ++v) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) s += st[v * 5 + c] * pp[v * 5 + c];
        const float a = __expf(s - mx) * inv;
        aw[(t * H + hh) * V + v] = a;
        if (tok < ntok) attw[((size_t)tok * H + hh) * V + v] = a;
      }
    }
  } else {
    for (int e = tid; e < VA_T * H * V; e += 256) {
      const int64_t tok = tok0 + e / (H * V);
      aw[e] = tok < ntok ? attw[(size_t)tok0 * H * V + e] : 0.f;
      da[e] = 0.f;
    }
  }
  __syncthreads();
  // ---- channel work: thread owns 4-channel chunks ch = tid + 256*k -----------------------------------
  const int nchunk = D / 4;
  for (int ch = tid; ch < nchunk; ch += 256) {
    const int i0 = ch * 4;
    const int hh = i0 / dh;
    if (!BWD) {
      float acc[VA_T][4];
#pragma unroll
      for (int t = 0; t < VA_T; ++t) { acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f; }
      for (int v = 0; v < V; ++v) {
        f32x4 g[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) g[c] = *reinterpret_cast<const f32x4*>(gtab + ((size_t)(v * 5 + c)) * D + i0);
#pragma unroll
        for (int t = 0; t < VA_T; ++t) {
          const float* p = pt + (t * V + v) * 5;
          const float a = aw[(t * H + hh) * V + v];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float val = g[0][j] * p[0] + g[1][j] * p[1] + g[2][j] * p[2] + g[3][j] * p[3] + g[4][j];
            acc[t][j] += a * val;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < VA_T; ++t) {
        const int64_t tok = tok0 + t;
        if (tok < ntok) {
          u32x2 o; o[0] = pack_bf2(acc[t][0], acc[t][1]); o[1] = pack_bf2(acc[t][2], acc[t][3]);
          *reinterpret_cast<u32x2*>(z + (size_t)tok * D + i0) = o;
        }
      }
    } else {
      float dzv[VA_T][4];
#pragma unroll
      for (int t = 0; t < VA_T; ++t) {
        const int64_t tok = tok0 + t;
        if (tok < ntok) {
          const u32x2 r = *reinterpret_cast<const u32x2*>(dz + (size_t)tok * D + i0);
          dzv[t][0] = bf2f((bf16_t)(r[0] & 0xffff)); dzv[t][1] = bf2f((bf16_t)(r[0] >> 16));
          dzv[t][2] = bf2f((bf16_t)(r[1] & 0xffff)); dzv[t][3] = bf2f((bf16_t)(r[1] >> 16));
        } else { dzv[t][0] = dzv[t][1] = dzv[t][2] = dzv[t][3] = 0.f; }
      }
      for (int v = 0; v < V; ++v) {
        f32x4 g[5];
        float dg[5][4];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          g[c] = *reinterpret_cast<const f32x4*>(gtab + ((size_t)(v * 5 + c)) * D + i0);
          dg[c][0] = dg[c][1] = dg[c][2] = dg[c][3] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < VA_T; ++t) {
          const float* p = pt + (t * V + v) * 5;
          const float a = aw[(t * H + hh) * V + v];
          float dav = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float val = g[0][j] * p[0] + g[1][j] * p[1] + g[2][j] * p[2] + g[3][j] * p[3] + g[4][j];
            dav += dzv[t][j] * val;
            const float ad = a * dzv[t][j];
#pragma unroll
            for (int c = 0; c < 5; ++c) dg[c][j] += ad * p[c];
          }
          atomicAdd(&da[(t * H + hh) * V + v], dav);   // LDS atomic: reduce over the head's channels
        }
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j) atomicAdd(dgtab + ((size_t)(v * 5 + c)) * D + i0 + j, dg[c][j]);
      }
    }
  }
  if (BWD) {
    __syncthreads();
    // softmax backward per (token, head): ds_v = a_v (da_v - sum_u a_u da_u); result overwrites da
    for (int e = tid; e < VA_T * H; e += 256) {
      float dot = 0.f;
      for (int v = 0; v < V; ++v) dot += aw[e * V + v] * da[e * V + v];
      for (int v = 0; v < V; ++v) da[e * V + v] = aw[e * V + v] * (da[e * V + v] - dot);
    }
    __syncthreads();
    // dstab[h][v][c] += sum_t ds[t][h][v] * pt[t][v][c]
    for (int e = tid; e < H * V * 5; e += 256) {
      const int c = e % 5, v = (e / 5) % V, hh = e / (5 * V);
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < VA_T; ++t) s += da[(t * H + hh) * V + v] * pt[(t * V + v) * 5 + c];
      atomicAdd(dstab + e, s);
    }
  }
}

}  // namespace

static int va_check(int B, int V, int h, int w, int H, int D) {
  if (B <= 0 || V <= 0 || V > VA_MAXV || H <= 0 || H > VA_MAXH || D <= 0) return O2_ERR_ARG;
  if ((h & 1) || (w & 1) || (D % 4) || (D % H) || ((D / H) % 4)) return O2_ERR_ARG;
  return O2_OK;
}

extern "C" int orbit2_varagg_fwd(const float* x, const float* stab, const float* gtab, void* z, float* attw, int B,
                                 int V, int h, int w, int H, int D, void* stream) {
  if (!x || !stab || !gtab || !z || !attw) return O2_ERR_ARG;
  int rc = va_check(B, V, h, w, H, D);
  if (rc) return rc;
  const int64_t ntok = (int64_t)B * (h / 2) * (w / 2);
  const size_t shm = sizeof(float) * (size_t)(VA_T * V * 5 + 2 * VA_T * H * V);
  hipLaunchKernelGGL(varagg_kernel<false>, dim3((unsigned)((ntok + VA_T - 1) / VA_T)), dim3(256), shm,
                     (hipStream_t)stream, x, stab, gtab, (bf16_t*)z, attw, (const bf16_t*)nullptr, (float*)nullptr,
                     (float*)nullptr, B, V, h, w, H, D);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

extern "C" int orbit2_varagg_bwd(const float* x, const float* gtab, const float* attw, const void* dz, float* dstab,
                                 float* dgtab, int B, int V, int h, int w, int H, int D, void* stream) {
  if (!x || !gtab || !attw || !dz || !dstab || !dgtab) return O2_ERR_ARG;
  int rc = va_check(B, V, h, w, H, D);
  if (rc) return rc;
  const int64_t ntok = (int64_t)B * (h / 2) * (w / 2);
  const size_t shm = sizeof(float) * (size_t)(VA_T * V * 5 + 2 * VA_T * H * V);
  hipLaunchKernelGGL(varagg_kernel<true>, dim3((unsigned)((ntok + VA_T - 1) / VA_T)), dim3(256), shm,
                     (hipStream_t)stream, x, (const float*)nullptr, gtab, (bf16_t*)nullptr, (float*)attw,
                     (const bf16_t*)dz, dstab, dgtab, B, V, h, w, H, D);
  O2_CHECK_LAUNCH();
  return O2_OK;
}
