/* liborbit2_hip.so -- C ABI of the MI355X-native (gfx950) hot path of ORBIT-2's
 * intermediate_downscaling training step (Res_Slim_ViT forward/backward + losses + AdamW).
 *
 * The reference (/root/reference) is 100 % Python and has no FFI seam; its operator seams are
 * the nn.Module forward()s and the FusedAttn switch.  Each entry point below names the reference
 * interface (file:line, relative to /root/reference) whose arithmetic it replaces.
 *
 * Conventions: every pointer is caller-owned DEVICE memory, row-major contiguous
 * unless a leading dimension is given; bf16 tensors are raw uint16; no allocation, no sync;
 * asynchronous on `stream` (a hipStream_t passed as void*); returns 0 on success, <0 on error
 * (never throws).  RNG = counter-based hash of (seed, element index) -- the caller advances
 * `seed` per call site and per step.
 *
 * State: the library keeps NO host-side mutable state and exactly ONE piece of device-side
 * mutable state, the 64-bit seed salt written by orbit2_seed_salt() (see there): one value per
 * device per process, 0 unless set, read by every seeded kernel on every stream.  Everything else
 * a call touches is passed in.  Calls are re-entrant across streams and threads as long as no
 * call that WRITES the salt is in flight concurrently with seeded kernels of another stream whose
 * forward and backward must agree (the salt is read at kernel run time, not at launch).
 */
#ifndef ORBIT2_HIP_H
#define ORBIT2_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORBIT2_ABI_VERSION 6
int orbit2_abi_version(void);

/* ---- bf16 MFMA GEMM with fused epilogue ------------------------------------------------
 * C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ),   fp32 accumulate.
 * a_kc=1: A stored [M][lda] (K contiguous);  a_kc=0: A stored [K][lda] (M contiguous).
 * b_kc=1: B stored [N][ldb] (K contiguous);  b_kc=0: B stored [K][ldb] (N contiguous).
 * Replaces every nn.Linear on the path: attention.py:36,40,50,81; mlp.py:50,54,63,67;
 * res_slimvit.py:115-120,326 (head); var_agg.proj attention.py:129,177 -- forward (a_kc=b_kc=1),
 * input-gradient (a_kc=1,b_kc=0) and weight-gradient (a_kc=b_kc=0) forms.
 * Epilogue order: +bias -> *colscale (columns n < colscale_n) -> save_pre -> GELU -> [+residual if res_first] -> dropout
 *   [-> save_dact] -> *gelu'(dgelu_pre) -> *mul -> *rowscale[m / rows_per_scale] -> [+residual] -> C = beta*C + v.
 * Requirements: N % 8 == 0; M % 8 == 0 unless a_kc (any M then); K % 8 == 0 if an operand is K-contiguous, any K when
 * both are K-strided (the weight-gradient form: K = tokens); lda/ldb/ldc % 8 == 0, 16-byte aligned bases. */
typedef struct {
  const void* A; const void* B; void* C;
  int M, N, K;
  int lda, ldb, ldc;
  int a_kc, b_kc;
  const void* bias;        /* bf16 [N] or NULL */
  int act;                 /* 0 none, 1 GELU(erf)  (nn.GELU default, mlp.py:64), 2 ReLU (VGG16 convs of LPIPS) */
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
  int tile_hint;           /* 0 = auto; tests / tuning: 128 = 128-tile kernel on 128 x 128 tiles, 64 = the same kernel on 64-row tiles
                              (A K-contiguous; auto takes them when the 128 x 128 grid is under two tiles per CU), 256 = 8-wave 8-phase kernel (K % 64 == 0),
                              260 = 4-wave kernel (M, N % 256 == 0, K % 64 == 0), 262 = 4-wave kernel with the runtime epilogue in
                              place of the compile-time kinds */
  int colscale_n;          /* columns n < colscale_n (a multiple of 8; 0 = none) are multiplied by colscale in fp32 right after */
  float colscale;          /*   the bias: the qkv Linear stores q * log2(e)/sqrt(d) (attention.py:50,54: q * scale), rounded ONCE */
  void* save_dact;         /* int16 [M][ldc] or NULL (needs act == 1): GELU'(pre) x (kept ? 1 / (1 - p) : 0) of THIS element as signed
                              fixed point with 14 fraction bits, range [-2, 2), saturating (at drop_p = 0.1 the factor lies in
                              [-0.15, 1.26]: 3e-5 absolute, where bf16 would give 4e-3; drop_p >= 0.434, where 1.13 / (1 - p)
                              reaches 2, is refused with O2_ERR_UNSUPPORTED: use save_pre / dgelu_pre there) -- what the backward multiplies the input gradient by (autograd of mlp.py:64-65), computed
                              here, where the pre-activation and the dropout decision are in registers, instead of GELU' + the
                              mask again in the backward */
  const void* mul;         /* int16 q14 [M][ldc] or NULL: multiply the result elementwise (the backward's use of a save_dact tensor) */
  float* colsum_ws;        /* fp32 [orbit2_gemm_bf16_colsum_rows(args)][N] or NULL (ABI 5): row t receives the column sums of the STORED
                              (bf16-rounded) output over rows 256 t .. 256 t + 255 -- the bias gradient of the layer below a GELU
                              (autograd of mlp.py:50,63) without a second pass over the 3 GB tensor: add the rows (orbit2_colsum on the
                              workspace).  Only the multiply-by-factor input gradient on whole tiles fills it: a call for which
                              orbit2_gemm_bf16_colsum_rows returns 0 is refused with O2_ERR_UNSUPPORTED when colsum_ws is set */
} orbit2_gemm_args;
int orbit2_gemm_bf16(const orbit2_gemm_args* args, void* stream);
int orbit2_gemm_bf16_colsum_rows(const orbit2_gemm_args* args);   /* 0: this call cannot fuse the column sums (colsum_ws must be NULL) */

/* n (<= ORBIT2_GEMM_MAX_GROUP) independent problems of ONE operand form (same a_kc, b_kc) in one launch (the 256x256
 * 8-phase kernel when every problem has K % 64 == 0, M, N >= 256 and the group fills the chip; the 128x128 kernel
 * otherwise): the partially filled last round of each problem is filled with the next one's tiles.  Used for
 * the four weight-gradient GEMMs of a Block (reference: autograd of attention.py:36,40 + mlp.py:50,54).
 * Round 6 (same ABI version: no signature or layout changed, the limit only grew from 8): up to 12 problems, so that a caller can
 * hand over the tiles beyond the group's last whole round of 256 as part-length problems over slices of the contraction (the
 * Python layer's balanced weight-gradient launch: 3 full problems + 2 x 4 quarter-length ones, partial products summed by
 * orbit2_batch_sum).  When every problem sweeps >= 512 K-tiles of 64 the workgroups of an XCD start their tiles together
 * (a bounded wait on a self-cleaning counter in a static device array: a pacing hint, never needed for correctness;
 * ORBIT2_W4_PACE = 0 / 1 / 2: off / on (default) / plus check points inside the sweep). */
#define ORBIT2_GEMM_MAX_GROUP 12
int orbit2_gemm_bf16_grouped(const orbit2_gemm_args* args, int n, void* stream);

/* small fp32 GEMM (parameter-table algebra of the folded variable aggregation):
 * C[M,N] = alpha * op(A) * op(B) + beta*C, row-major fp32; ta/tb: 0 = as stored, 1 = transposed. */
int orbit2_sgemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                     int ta, int tb, float alpha, float beta, void* stream);
/* same product with a caller-owned fp32 workspace: skinny problems (a few rows against a D x D weight) are split over K
 * into orbit2_sgemm_f32_ws_floats(M,N,K) / (M*N) slabs and combined deterministically; 0 floats = no split is planned */
int64_t orbit2_sgemm_f32_ws_floats(int M, int N, int K);
int orbit2_sgemm_f32_ws(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                        int ta, int tb, float alpha, float beta, float* ws, int64_t ws_floats, void* stream);

/* ---- LayerNorm (vit_blocks.py:46,63; res_slimvit.py:104,294): eps 1e-5, affine ------------ */
int orbit2_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                         int rows, int D, float eps, void* stream);
/* the same with a row pitch on y (ldy >= D elements, a multiple of 8; ABI 5): the output is the A operand of the next GEMM, and
 * rows a multiple of 8 KiB apart (D % 4096 == 0: interm_10b) put every row's k-offset on one memory channel */
int orbit2_layernorm_fwd_ld(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                            int rows, int D, int ldy, float eps, void* stream);
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
/* dqkv: bf16 [B, L, 3, H, d];  delta: fp32 workspace of orbit2_attn_bwd_ws_floats(B, L, H) floats (two per-row statistics
 * tables, -lse log2(e) and -rowsum(dO o O) / dropout scale, padded per (b, h): the dK / dV kernels copy their tiles of it into
 * LDS by LDS-DMA) */
int64_t orbit2_attn_bwd_ws_floats(int B, int L, int H);
int orbit2_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                    void* dqkv, int B, int L, int H, int d, float drop_p, uint64_t seed, void* stream);
/* The same two entries with the kernel variant as an ARGUMENT (A/B timing and the bit-equality tests of the fused
 * d = 128 dK+dV pass): flags = 0 is what orbit2_attn_fwd / _bwd run.  Nothing on the launch path reads the environment
 * or any other process-global switch. */
#define ORBIT2_ATTN_4WAVES 1     /* 4-wave / 128-row workgroups (round-1 geometry) instead of 8-wave / 256-row ones */
#define ORBIT2_ATTN_SPLIT_DKV 2  /* d = 128: dK and dV as two passes instead of the fused one */
/* The q third of qkv already holds q * log2(e)/sqrt(d) (written so by the qkv GEMM's colscale epilogue: ONE rounding to
 * bf16, products with k exact, like the reference's fp32 scaling of q k^T).  Without the flag the kernels multiply their
 * register-resident operand (q, or k in the dK/dV pass) by that factor themselves and round it to bf16 a second time
 * (relative 2^-9 per element of that operand: harmless at ordinary score magnitudes, ~1e-2 of the output when scores reach
 * tens of nats).  dqkv is in both cases the gradient with respect to the UNSCALED q, k, v. */
#define ORBIT2_ATTN_Q_PRESCALED 4
/* Forward and the dQ pass of the backward at d = 128 with ORBIT2_ATTN_Q_PRESCALED and L a multiple of 256 (<= 16384) run the
 * generated one-wave-per-SIMD kernels (csrc/attn_fwd_asm.h, attn_dq_asm.h; tools/gen_attn_fwd.py, gen_attn_dq.py); this flag
 * keeps the compiler-scheduled kernels instead (A/B timing, tests). */
#define ORBIT2_ATTN_NO_W4 8
int orbit2_attn_fwd_ex(const void* qkv, void* out, float* lse, int B, int L, int H, int d, float drop_p,
                       uint64_t seed, int flags, void* stream);
int orbit2_attn_bwd_ex(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                       void* dqkv, int B, int L, int H, int d, float drop_p, uint64_t seed, int flags, void* stream);
/* The same two with token-row pitches (elements, multiples of 8; ABI 5): qkv[b, l] and dqkv[b, l] start at (b * L + l) * ldq
 * (ldq >= 3 * H * d), out[b, l] at (b * L + l) * ldo (ldo >= H * d) -- these tensors are GEMM operands on the other side, and rows
 * a multiple of 8 KiB apart put every row's k-offset on one memory channel (see orbit2_layernorm_fwd_ld).  dout stays [B, L, H, d]. */
int orbit2_attn_fwd_ld(const void* qkv, void* out, float* lse, int B, int L, int H, int d, float drop_p,
                       uint64_t seed, int flags, int ldq, int ldo, void* stream);
int orbit2_attn_bwd_ld(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                       void* dqkv, int B, int L, int H, int d, float drop_p, uint64_t seed, int flags, int ldq, int ldo, void* stream);

/* ---- folded patch-embed + variable aggregation (res_slimvit.py:250-265, 205-230;
 *      patch_embed.py:44-52; attention.py:132-176) ---------------------------------------------
 * x: fp32 [B, V, h, w]; stab: fp32 [H][V][5] score table; gtab: fp32 [V][5][D] value table
 * (both functions of the weights only);  z: bf16 [B*L, D] (input of var_agg.proj);
 * attw: fp32 [B*L, H, V] softmax weights saved for backward.  patch must be 2. */
int orbit2_varagg_fwd(const float* x, const float* stab, const float* gtab, void* z, float* attw, int B, int V,
                      int h, int w, int H, int D, void* stream);
/* dstab/dgtab are ACCUMULATED into (caller zeroes them).  ws: fp32 workspace of orbit2_varagg_bwd_ws_floats(...) floats (ABI 4):
 * every (head, token range) workgroup stores its partial tables in its own slab and the ranges are added in a fixed order --
 * no float atomics, bitwise reproducible. */
int64_t orbit2_varagg_bwd_ws_floats(int B, int V, int h, int w, int H, int D);
int orbit2_varagg_bwd(const float* x, const float* gtab, const float* attw, const void* dz, float* dstab,
                      float* dgtab, int B, int V, int h, int w, int H, int D, float* ws, void* stream);
/* The per-variable rows both tables are built from (ABI 6): cmat[(v, c)][D], c = 0..3 = the 2x2 patch-embed weight of variable
 * ids[v] transposed ([D][4] -> 4 rows), c = 4 = its bias + var_embed[ids[v]]  (res_slimvit.py:64-66 PatchEmbed x V, :182-201
 * var-embed gather, :251-262).  The V_total per-variable parameters are addressed as base + index * stride (elements): a caller
 * whose parameters lie at a uniform pitch (a flat parameter buffer) builds the rows in one launch; _scatter is the transpose and
 * ACCUMULATES into the gradient buffers at the same pitches (ids distinct). */
int orbit2_tables_gather(const float* w_base, int64_t w_stride, const float* b_base, int64_t b_stride, const float* var_embed,
                         const int* ids, float* cmat, int V, int D, void* stream);
int orbit2_tables_scatter(const float* dcmat, float* dw_base, int64_t w_stride, float* db_base, int64_t b_stride,
                          float* dvar_embed, const int* ids, int V, int D, void* stream);
/* 1 if orbit2_varagg_bwd sums the table gradients in a fixed order for this shape (bitwise reproducible), 0 if it takes the
 * scalar fallback that accumulates them with fp32 atomics (head dim not 64 / 128 / 256, 5 V > 128, ORBIT2_VARAGG_SCALAR set):
 * callers that keep replicas of the tables in lock-step without exchanging gradients need to know (dist/tp.py ReplicaGuard) */
int orbit2_varagg_bwd_is_fixed_order(int B, int V, int h, int w, int H, int D);

/* ---- elementwise / reductions ------------------------------------------------------------- */
/* dym = dy * dropmask * rowscale (backward of the dropout/DropPath epilogue); dym may alias dy */
int orbit2_dropout_bwd(const void* dy, void* dym, int M, int N, float drop_p, uint64_t seed, const float* rowscale,
                       int rows_per_scale, void* stream);
/* the same, fused with colsum_out[N] = beta*colsum_out + sum_m dym[m][n] (the bias gradient of the Linear whose output
 * gradient dym is: reference autograd of attention.py:40,80-82 / mlp.py:54,66-68): dym is not re-read from HBM; result
 * bit-identical to orbit2_dropout_bwd + orbit2_colsum.  ws: fp32 >= orbit2_colsum_ws_floats(M, N) */
int orbit2_dropout_bwd_colsum(const void* dy, void* dym, int M, int N, float drop_p, uint64_t seed, const float* rowscale,
                              int rows_per_scale, void* colsum_out, int out_fp32, float beta, float* ws, int ws_floats,
                              void* stream);
/* y = residual + rowscale[m/rows_per_scale] * dropout(x + addend[m % res_mod]) (bf16; every term optional): the part of
 * the GEMM epilogue that has to wait for the all-reduce of tensor-parallel partial products (row-parallel proj / fc2,
 * reference attention.py:81-85, mlp.py:66-71).  Same mask hash as the epilogue; y may alias x. */
int orbit2_post_reduce(const void* x, const void* addend, int res_mod, const void* residual, void* y, int M, int N,
                       float drop_p, uint64_t seed, const float* rowscale, int rows_per_scale, void* stream);
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

/* ---- position-embedding table of a step (ABI 6) ------------------------------------------------ */
/* out[nh*nw][D] = bicubic(pe seen as [oh][ow][D]) + sw[:] * res + sb[:]  (fp32; sw, sb both NULL: plain re-grid).
 * Replaces components/pos_embed.py:103-138 interpolate_pos_embed_on_the_fly (F.interpolate mode="bicubic",
 * align_corners=False on the [1, L0, D] table, taken only when oh != nh -- otherwise the table is used as it is) and the
 * Linear(1, D) resolution embedding res_slimvit.py:62,277-281.  D % 4 == 0, sides <= 2048. */
int orbit2_posembed_fwd(const float* pe, const float* sw, const float* sb, float res, float* out, int oh, int ow, int nh,
                        int nw, int D, void* stream);
/* transpose of the re-grid (autograd of the call above w.r.t. pe): dpe[oh*ow][D] from dout[nh*nw][D]; fixed summation
 * order, no atomics.  (d sb = column sums of dout, d sw = res * d sb: orbit2_colsum.) */
int orbit2_posembed_bwd(const float* dout, float* dpe, int oh, int ow, int nh, int nw, int D, void* stream);

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
 * din (may be NULL) fp32 [B,Cin,H,W]; dweight/dbias fp32, ACCUMULATED into (caller zeroes).
 * ws: fp32 workspace of orbit2_conv3x3_bwd_ws_floats(...) floats (ABI 4): one slab of Cout*Cin*9 + Cout partial sums per
 * 16x16-pixel tile, added in a fixed order (no float atomics: the gradients are bitwise reproducible). */
int64_t orbit2_conv3x3_bwd_ws_floats(int B, int Cin, int Cout, int H, int W);
int orbit2_conv3x3_bwd(const float* dout, const float* in, const int* chan_idx, int in_ctotal, const float* weight,
                       const float* pre, float* din, float* dweight, float* dbias, int B, int Cin, int Cout, int H,
                       int W, int mode, int r, float* ws, void* stream);
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

/* evaluation metrics (metrics/functional.py:219-324 mae / rmse / acc / pearson / mean_bias): out[b][c][12] (double),
 * with a = pred - clim, b = target - clim (clim fp32 [C][H][W] or NULL; target: top-left crop), w = lat_w[y] or 1:
 * {sum a, sum b, sum a^2, sum b^2, sum ab, sum w(a-b)^2, sum w|a-b|, sum wa, sum wb, sum w ab, sum w a^2, sum w b^2} */
int orbit2_eval_moments(const float* pred, const float* target, int Ht, int Wt, const float* lat_w, const float* clim,
                        double* out, int B, int C, int H, int W, void* stream);

/* ---- perceptual loss = L1 + 0.5 * mean_b LPIPS-VGG16 (metrics/functional.py:17-33, metrics.py:119-187) ------
 * Feature maps are NHWC bf16, so each 3x3 VGG convolution is im2col + orbit2_gemm_bf16 (bias, act = 2) forward and
 * orbit2_gemm_bf16 + col2im backward (input gradient only: LPIPS weights are frozen, metrics.py:127-128).
 * Tap order of a 3x3 window: t = ky*3 + kx, offsets (ky-1, kx-1), zero padding.  C % 8 == 0. */
/* col[p][t][c] = x[p + off_t][c];  x: [N][H][W][C], col: [N*H*W][9*C] */
int orbit2_im2col3x3(const void* x, void* col, int N, int H, int W, int C, void* stream);
/* g[p][c] = sum_t dcol[p - off_t][t][c]; with act != NULL: out = (g + tapg) * (act > 0)  (ReLU backward of the layer
 * that produced act, plus the LPIPS tap gradient at that layer; tapg may be NULL) */
int orbit2_col2im3x3(const void* dcol, const void* act, const void* tapg, void* out, int N, int H, int W, int C,
                     void* stream);
/* 2x2 / stride 2 max-pool (torchvision VGG16 features 4, 9, 16, 23); backward routes to the first maximum of the
 * window (ATen's index rule) and fuses the ReLU mask of x and the tap gradient like col2im */
int orbit2_maxpool2_fwd(const void* x, void* y, int N, int H, int W, int C, void* stream);
int orbit2_maxpool2_bwd(const void* g, const void* x, const void* tapg, void* dz, int N, int H, int W, int C,
                        void* stream);
/* first convolution 3 -> 64 (+ReLU) straight from the NCHW fp32 image with the LPIPS ScalingLayer fused;
 * w1: fp32 [(t*3 + ci)][64], b1: fp32 [64]; out: [N][H][W][64] bf16 */
int orbit2_lpips_conv1_fwd(const float* img, const float* w1, const float* b1, void* out, int N, int H, int W,
                           void* stream);
/* dimg (NCHW fp32) = conv1 input gradient / scale + l1_coef * gscale[0] * sign(pred - target)   (the L1 term of the loss).
 * gscale (here and in orbit2_lpips_tap_bwd): device pointer to the scalar upstream gradient of the loss (loss scale included)
 * or NULL for 1 -- the backward never reads it on the host, so the loss can be captured in a hipGraph. */
int orbit2_lpips_conv1_bwd(const void* dz, const float* w1, const float* pred, const float* target, float l1_coef,
                           const float* gscale, float* dimg, int N, int H, int W, void* stream);
/* LPIPS head of one tap.  feats: [2B][HW][C] bf16, images 0..B-1 = prediction, B..2B-1 = target; lin: fp32 [C].
 * fwd: val[b] += mean_px sum_c lin_c (f0_c/(|f0|+1e-10) - f1_c/(|f1|+1e-10))^2.   C in {64,128,256,512}.
 * bwd: gout[b][px][c] = coef * gscale[0] * d(sum_c ...)/d f0_c * (f0_c > 0)  -- the gradient w.r.t. the tap's PRE-ReLU output
 * (bf16; a pixel whose prediction features are all zero gets 0 where autograd of sqrt at 0 would produce NaN)
 * ws (fwd, ABI 4): fp32 workspace of orbit2_lpips_tap_ws_floats(B, HW, C) floats -- per-block partial sums added in a fixed order */
int64_t orbit2_lpips_tap_ws_floats(int B, int HW, int C);
int orbit2_lpips_tap_fwd(const void* feats, const float* lin, float* val, int B, int HW, int C, float* ws, void* stream);
int orbit2_lpips_tap_bwd(const void* feats, const float* lin, void* gout, float coef, const float* gscale, int B, int HW, int C,
                         void* stream);
/* out[0] += mean |a - b|   (F.l1_loss, metrics/functional.py:30); ws: orbit2_l1_mean_ws_floats(n) floats (fixed-order sum, ABI 4) */
int64_t orbit2_l1_mean_ws_floats(int64_t n);
int orbit2_l1_mean(const float* a, const float* b, float* out, int64_t n, float* ws, void* stream);

/* ---- optimizer (utils/loaders.py:398-399 AdamW; ShardedGradScaler :732-742) ------------------ */
/* flat fused AdamW over n elements: fp32 master p/m/v, gradient g (bf16 or fp32) multiplied by
 * grad_scale; writes the bf16 compute copy p16 (may be NULL).  Skips everything when *found_inf != 0. */
int orbit2_adamw(float* p, float* m, float* v, const void* g, int g_fp32, void* p16, int64_t n, float lr,
                 float beta1, float beta2, float eps, float wd, float bc1, float bc2, float grad_scale,
                 const float* found_inf, void* stream);
/* found_inf[0] = 1 if any element is inf/nan (never cleared here) */
int orbit2_check_finite(const void* g, int g_fp32, int64_t n, float* found_inf, void* stream);

/* Device-side seed salt -- the library's only device-global mutable state: every seeded kernel (GEMM-epilogue dropout,
 * attention dropout, dropout backward, DropPath scales) xors it into the seed argument.  add = 0 sets it, add = 1 advances it
 * by `value`; stream-ordered (a one-thread kernel per library module).  0 by default, i.e. the seeds are used as passed.
 * Purpose: a training step captured in a hipGraph begins with orbit2_seed_salt(odd constant, 1, stream), so each replay draws
 * new masks (seeds are kernel arguments, frozen at capture).
 * Consequences of it being per device and not per engine: (1) the salt must not change between a step's forward and the
 * backward that regenerates its masks -- within one stream the stream order guarantees it; two engines stepping CONCURRENTLY on
 * different streams of one device must not both use salted (graph-captured) steps; (2) engines that take turns on one device
 * share the counter: each one's mask sequence then depends on how many replays the others ran (still fresh masks every step,
 * still forward/backward-consistent, but not reproducible per engine); (3) eager steps that follow replays see the salt the
 * last replay left (set it back with add = 0 when bit-reproducing an eager run).  One process per GPU with one training
 * engine -- the reference's layout (examples/intermediate_downscaling.py:161-262) -- meets none of these cases. */
int orbit2_seed_salt(uint64_t value, int add, void* stream);

/* hardware self-test of the MFMA / LDS-transpose / LDS-DMA layouts the kernels assume; returns a
 * bitmask of failed checks in result[0] (0 = all good). */
int orbit2_selftest(int* result, void* stream);
/* diagnostic: sweep `bytes` of buf with 16-byte loads from `blocks` workgroups, `inflight` (1, 4 or 8) loads per lane at a time --
 * the calibration streams of the memory-side latency probe (tools/mall_probe.py, bench.py --mall-probe); sink: one float */
int orbit2_probe_read(const void* buf, int64_t bytes, int blocks, int inflight, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif
