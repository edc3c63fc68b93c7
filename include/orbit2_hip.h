/* liborbit2_hip.so -- C ABI of the MI355X-native (gfx950) hot path of ORBIT-2's
 * intermediate_downscaling training step (Res_Slim_ViT forward/backward + losses + AdamW).
 *
 * The reference (/root/reference) is 100 % Python and has no FFI seam; its operator seams are
 * the nn.Module forward()s and the FusedAttn switch.  Each entry point below names the reference
 * interface (file:line, relative to /root/reference) whose arithmetic it replaces.
 *
 * Conventions: every pointer is caller-owned DEVICE memory, row-major contiguous
 * unless a leading dimension is given; bf16 tensors are raw uint16; no allocation, no sync, no
 * global mutable state inside; asynchronous on `stream` (a hipStream_t passed as void*);
 * returns 0 on success, <0 on error (never throws).  RNG = counter-based hash of (seed, element
 * index) -- the caller advances `seed` per call site and per step.
 */
#ifndef ORBIT2_HIP_H
#define ORBIT2_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORBIT2_ABI_VERSION 1
int orbit2_abi_version(void);

/* ---- bf16 MFMA GEMM with fused epilogue ------------------------------------------------
 * C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ),   fp32 accumulate.
 * a_kc=1: A stored [M][lda] (K contiguous);  a_kc=0: A stored [K][lda] (M contiguous).
 * b_kc=1: B stored [N][ldb] (K contiguous);  b_kc=0: B stored [K][ldb] (N contiguous).
 * Replaces every nn.Linear on the path: attention.py:36,40,50,81; mlp.py:50,54,63,67;
 * res_slimvit.py:115-120,326 (head); var_agg.proj attention.py:129,177 -- forward (a_kc=b_kc=1),
 * input-gradient (a_kc=1,b_kc=0) and weight-gradient (a_kc=b_kc=0) forms.
 * Epilogue order: +bias -> save_pre -> GELU -> [+residual if res_first] -> dropout ->
 *   *gelu'(dgelu_pre) -> *rowscale[m / rows_per_scale] -> [+residual] -> C = beta*C + v.
 * Requirements: K % 64 == 0, N % 8 == 0, M % 8 == 0, lda/ldb/ldc % 8 == 0, 16-byte aligned bases. */
typedef struct {
  const void* A; const void* B; void* C;
  int M, N, K;
  int lda, ldb, ldc;
  int a_kc, b_kc;
  const void* bias;        /* bf16 [N] or NULL */
  int act;                 /* 0 none, 1 GELU(erf)  (nn.GELU default, mlp.py:64) */
  void* save_pre;          /* bf16 [M][ldc] pre-activation copy, or NULL */
  const void* dgelu_pre;   /* bf16 [M][ldc]: multiply by GELU'(pre), or NULL */
  float drop_p;            /* nn.Dropout on the output element (attention.py:82, mlp.py:65,68) */
  uint64_t seed;
  const float* rowscale;   /* DropPath (vit_blocks.py:78-79): fp32 [M / rows_per_scale], or NULL */
  int rows_per_scale;
  const void* residual;    /* bf16 [res_mod][ldr] added at row (m % res_mod), or NULL */
  int ldr, res_mod, res_first;
  int out_fp32;            /* 0: C is bf16, 1: C is fp32 */
  float beta;              /* C = beta*C + result (gradient accumulation) */
  int tile_hint;           /* 0 = auto, 128 or 256 = force that kernel (tests / tuning) */
} orbit2_gemm_args;
int orbit2_gemm_bf16(const orbit2_gemm_args* args, void* stream);

/* n (<= ORBIT2_GEMM_MAX_GROUP) independent problems of ONE operand form (same a_kc, b_kc) in one launch of the
 * 128x128 kernel: the partially filled last round of each problem is filled with the next one's tiles.  Used for
 * the four weight-gradient GEMMs of a Block (reference: autograd of attention.py:36,40 + mlp.py:50,54). */
#define ORBIT2_GEMM_MAX_GROUP 8
int orbit2_gemm_bf16_grouped(const orbit2_gemm_args* args, int n, void* stream);

/* small fp32 GEMM (parameter-table algebra of the folded variable aggregation):
 * C[M,N] = alpha * op(A) * op(B) + beta*C, row-major fp32; ta/tb: 0 = as stored, 1 = transposed. */
int orbit2_sgemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                     int ta, int tb, float alpha, float beta, void* stream);

/* ---- LayerNorm (vit_blocks.py:46,63; res_slimvit.py:104,294): eps 1e-5, affine ------------ */
int orbit2_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                         int rows, int D, float eps, void* stream);
/* dx = LN'(dy) [+ dres];  dgamma/dbeta: bf16 or fp32 [D] (beta_acc accumulates).  ws: fp32 >= 2*D*nblk */
int orbit2_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                         const void* dres, void* dx, void* dgamma, void* dbeta, int grads_fp32, float beta_acc,
                         float* ws, int ws_floats, int rows, int D, void* stream);
int orbit2_layernorm_bwd_ws_floats(int rows, int D);

/* ---- multi-head self-attention core (attention.py:54-78): softmax(q k^T / sqrt(d)) v --------
 * qkv: bf16 [B, L, 3, H, d] (the qkv Linear output as stored, no permute copies);
 * out: bf16 [B, L, H, d]; lse: fp32 [B, H, L].  d in {64, 128, 256}; any L >= 1 (ragged tails are masked).
 * drop_p: dropout on P (attention.py:57,69,76). */
int orbit2_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int d, float drop_p,
                    uint64_t seed, void* stream);
/* dqkv: bf16 [B, L, 3, H, d];  delta: fp32 workspace [B, H, L] */
int orbit2_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                    void* dqkv, int B, int L, int H, int d, float drop_p, uint64_t seed, void* stream);

/* ---- folded patch-embed + variable aggregation (res_slimvit.py:250-265, 205-230;
 *      patch_embed.py:44-52; attention.py:132-176) ---------------------------------------------
 * x: fp32 [B, V, h, w]; stab: fp32 [H][V][5] score table; gtab: fp32 [V][5][D] value table
 * (both functions of the weights only);  z: bf16 [B*L, D] (input of var_agg.proj);
 * attw: fp32 [B*L, H, V] softmax weights saved for backward.  patch must be 2. */
int orbit2_varagg_fwd(const float* x, const float* stab, const float* gtab, void* z, float* attw, int B, int V,
                      int h, int w, int H, int D, void* stream);
/* dstab/dgtab are ACCUMULATED into (caller zeroes them) */
int orbit2_varagg_bwd(const float* x, const float* gtab, const float* attw, const void* dz, float* dstab,
                      float* dgtab, int B, int V, int h, int w, int H, int D, void* stream);

/* ---- elementwise / reductions ------------------------------------------------------------- */
/* dym = dy * dropmask * rowscale (backward of the dropout/DropPath epilogue); dym may alias dy */
int orbit2_dropout_bwd(const void* dy, void* dym, int M, int N, float drop_p, uint64_t seed, const float* rowscale,
                       int rows_per_scale, void* stream);
/* out[N] = beta*out + sum_m x[m][n]  (bias gradients; sum over batch).  ws: fp32 >= colsum_ws_floats */
int orbit2_colsum(const void* x, int x_fp32, int M, int N, int ldx, void* out, int out_fp32, float beta, float* ws,
                  int ws_floats, void* stream);
int orbit2_colsum_ws_floats(int M, int N);
/* out[r][n] = sum_b x[b][r][n]  (pos_embed gradient over the batch, res_slimvit.py:273) */
int orbit2_batch_sum(const void* x, void* out, int B, int rows, int N, int out_fp32, float beta, void* stream);
/* DropPath (timm 0.9.2, vit_blocks.py:61,74): out[b] = 0 with prob p else 1/(1-p), from hash(seed, b) */
int orbit2_droppath_scales(float* out, int B, float p, uint64_t seed, void* stream);
/* dst[C][R] = src[R][C] (bf16): transposed compute copy of a weight so that dX = dY.W runs K-contiguous */
int orbit2_transpose_bf16(const void* src, void* dst, int R, int C, void* stream);
int orbit2_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int orbit2_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* y = a + b (bf16), used for pos_embed + spatial_embed table (res_slimvit.py:273-281) */
int orbit2_add_rowvec(const void* a, const void* vec, void* y, int rows, int N, void* stream);

/* ---- hi-res tail -------------------------------------------------------------------------- */
/* unpatchify (res_slimvit.py:167-179): t bf16 [B, L, C*(s*p)^2] -> img [B, C, h*s, w*s] (fp32) */
int orbit2_unpatchify_fwd(const void* t, float* img, int B, int C, int h, int w, int p, int s, void* stream);
int orbit2_unpatchify_bwd(const float* dimg, void* dt, int B, int C, int h, int w, int p, int s, void* stream);
/* 3x3 conv, stride 1, zero pad 1 (res_slimvit.py:108,111,122).  in: fp32 [B,Cin,H,W] gathered through
 * chan_idx (NULL = identity; res_slimvit.py:237 channel gather); weight fp32 [Cout,Cin,3,3]; bias [Cout].
 * mode 0: out[B,Cout,H,W];  mode 1: GELU then PixelShuffle(r) -> out[B,Cout/r^2,H*r,W*r]
 * (pre-activation saved to `pre` [B,Cout,H,W] for backward);  addend: optional fp32 image
 * [B,Cout,Ha,Wa] whose top-left HxW crop is added (res_slimvit.py:333-336). */
int orbit2_conv3x3_fwd(const float* in, const int* chan_idx, int in_ctotal, const float* weight, const float* bias,
                       float* out, float* pre, const float* addend, int Ha, int Wa, int B, int Cin, int Cout,
                       int H, int W, int mode, int r, void* stream);
/* backward: dout is [B,Cout,H,W] (mode 0) or the shuffled [B,Cout/r^2,H*r,W*r] (mode 1, needs pre).
 * din (may be NULL) fp32 [B,Cin,H,W]; dweight/dbias fp32, ACCUMULATED into (caller zeroes). */
int orbit2_conv3x3_bwd(const float* dout, const float* in, const int* chan_idx, int in_ctotal, const float* weight,
                       const float* pre, float* din, float* dweight, float* dbias, int B, int Cin, int Cout, int H,
                       int W, int mode, int r, void* stream);
/* in-place clamp of one channel at 0 (examples/intermediate_downscaling.py:267-272) */
int orbit2_clamp_channel(float* img, int B, int C, int HW, int chan, void* stream);
int orbit2_clamp_channel_bwd(const float* img_clamped, float* dimg, int B, int C, int HW, int chan, void* stream);

/* ---- losses (metrics/functional.py:59-202): kind 0 = mse, 1 = bayesian_tv, 2 = image_gradient ------
 * pred fp32 [B,C,H,W]; target fp32 [B,C,Ht,Wt] (top-left crop used); lat_w fp32 [H] or NULL;
 * chan_w fp32 [C] or NULL.  out: fp32 [C+1] (per-channel means, aggregate mean). ws fp32 >= 2*B*C*64 */
int orbit2_loss_fwd(const float* pred, const float* target, int Ht, int Wt, const float* lat_w, const float* chan_w,
                    float* out, float* ws, int B, int C, int H, int W, int kind, void* stream);
/* dpred = gscale[0] * d(aggregate)/dpred */
int orbit2_loss_bwd(const float* pred, const float* target, int Ht, int Wt, const float* lat_w, const float* chan_w,
                    const float* gscale, float* dpred, int B, int C, int H, int W, int kind, void* stream);

/* ---- optimizer (utils/loaders.py:398-399 AdamW; ShardedGradScaler :732-742) ------------------ */
/* flat fused AdamW over n elements: fp32 master p/m/v, gradient g (bf16 or fp32) multiplied by
 * grad_scale; writes the bf16 compute copy p16 (may be NULL).  Skips everything when *found_inf != 0. */
int orbit2_adamw(float* p, float* m, float* v, const void* g, int g_fp32, void* p16, int64_t n, float lr,
                 float beta1, float beta2, float eps, float wd, float bc1, float bc2, float grad_scale,
                 const float* found_inf, void* stream);
/* found_inf[0] = 1 if any element is inf/nan (never cleared here) */
int orbit2_check_finite(const void* g, int g_fp32, int64_t n, float* found_inf, void* stream);

/* hardware self-test of the MFMA / LDS-transpose / LDS-DMA layouts the kernels assume; returns a
 * bitmask of failed checks in result[0] (0 = all good). */
int orbit2_selftest(int* result, void* stream);

#ifdef __cplusplus
}
#endif
#endif
