// Hardware self-test: checks, with exact small-integer data, every lane<->element map the kernels rely on
// (MI355X guide section 3 / T10).  bit 0: 16x16x32 MFMA maps, bit 1: 32x32x16 MFMA maps, bit 2: LDS-DMA is
// lane-linear, bit 3: ds_read_b64_tr_b16 block transpose, bit 4: accumulator-as-operand k order, bit 5:
// v_permlane16_swap row map (wide GEMM epilogue), bit 6: fast GELU/GELU' within 1.5e-6 of the erff forms.
#include "common.h"
#include "../../include/orbit2_hip.h"

namespace {

__device__ __forceinline__ short i2bf(int v) { return (short)f2bf((float)v); }

__global__ __launch_bounds__(64) void selftest_kernel(int* result, const bf16_t* scratch_in) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2048];
  const int l = threadIdx.x;
  int fail = 0;
  // asymmetric integer matrices: A[i][k] = (3i + 5k) % 7 - 3, B[k][j] = (2k + 7j) % 5 - 2
  auto Af = [](int i, int k) { return (3 * i + 5 * k) % 7 - 3; };
  auto Bf = [](int k, int j) { return (2 * k + 7 * j) % 5 - 2; };
  {  // ---- 16x16x32
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = i2bf(Af(l & 15, 8 * (l >> 4) + j)); b[j] = i2bf(Bf(8 * (l >> 4) + j, l & 15)); }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
      const int row = (l >> 4) * 4 + r, col = l & 15;
      int ref = 0;
      for (int k = 0; k < 32; ++k) ref += Af(row, k) * Bf(k, col);
      if ((int)c[r] != ref) fail |= 1;
    }
  }
  f32x16 x32;
  {  // ---- 32x32x16
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = i2bf(Af(l & 31, 8 * (l >> 5) + j)); b[j] = i2bf(Bf(8 * (l >> 5) + j, l & 31)); }
    for (int r = 0; r < 16; ++r) x32[r] = 0.f;
    x32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x32, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      int ref = 0;
      for (int k = 0; k < 16; ++k) ref += Af(row, k) * Bf(k, col);
      if ((int)x32[r] != ref) fail |= 2;
    }
  }
  {  // ---- LDS-DMA lane-linear: lane l fetches 8 values scratch_in[(63-l)*8 ..] -> must land at lds[l*8 ..]
    glds16(scratch_in + (63 - l) * 8, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 8; ++j)
      if (lds[l * 8 + j] != (bf16_t)((63 - l) * 8 + j)) fail |= 4;
    __syncthreads();
  }
  {  // ---- transposed block read.  Tile T[row][col], 32 rows x 32 cols of u16 = row*64 + col, row stride 64 B
    for (int e = l; e < 32 * 32; e += 64) lds[e] = (bf16_t)((e >> 5) * 64 + (e & 31));
    __syncthreads();
    const int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    // group g reads the 4x16 block at rows 8g..8g+3, cols 16*(g&1)..+15
    const bf16x4 v = lds_tr4(reinterpret_cast<const char*>(lds) + (8 * g + q) * 64 + (16 * (g & 1) + 4 * p) * 2);
    for (int e = 0; e < 4; ++e)
      if ((unsigned short)v[e] != (unsigned short)((8 * g + e) * 64 + 16 * (g & 1) + i)) fail |= 8;
    __syncthreads();
  }
  {  // ---- accumulator as the next MFMA's B operand: Y = A2 . X, X = x32 (32x32, exact small ints)
    // A2[i][k] = (i + 2k) % 3 - 1 ; k-step s uses X rows 16s + 8(j>>2) + 4h + (j&3) for element j
    auto A2 = [](int i, int k) { return (i + 2 * k) % 3 - 1; };
    f32x16 y;
    for (int r = 0; r < 16; ++r) y[r] = 0.f;
    const int h = l >> 5;
    for (int s = 0; s < 2; ++s) {
      bf16x8 xa, xb;
      for (int j = 0; j < 8; ++j) {
        xb[j] = (short)f2bf(x32[8 * s + j]);
        xa[j] = i2bf(A2(l & 31, 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)));
      }
      y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, y, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      int ref = 0;
      for (int k = 0; k < 32; ++k) {
        int xk = 0;
        for (int kk = 0; kk < 16; ++kk) xk += Af(k, kk) * Bf(kk, col);
        ref += A2(row, k) * xk;
      }
      if ((int)y[r] != ref) fail |= 16;
    }
  }
  {  // ---- v_permlane16_swap: r0 = [a row0, b row0, a row2, b row2], r1 = [a row1, b row1, a row3, b row3]
    const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)l, 100u + (unsigned)l, false, false);
    const int r4 = l >> 4, i = l & 15;
    const unsigned e0 = ((r4 & 1) ? 100u : 0u) + 16u * (unsigned)(r4 & 2) + (unsigned)i;
    if (sw[0] != e0 || sw[1] != e0 + 16u) fail |= 32;
  }
  {  // ---- fast GELU / GELU' (bf16 GEMM epilogues) against the erff forms on [-9, 9]
    for (int k = 0; k < 64; ++k) {
      const float x = -9.0f + (float)(l * 64 + k) * (18.0f / 4095.0f);
      if (fabsf(gelu_fast(x) - gelu_f(x)) > 1.5e-6f || fabsf(dgelu_fast(x) - dgelu_f(x)) > 1.5e-6f) fail |= 64;
    }
  }
  if (fail) atomicOr(result, fail);
}

__global__ void selftest_fill(bf16_t* s) { s[threadIdx.x + blockIdx.x * 256] = (bf16_t)(threadIdx.x + blockIdx.x * 256); }

}  // namespace

// result: int[1 + 256] device buffer (result[0] = failure mask; the rest is scratch)
extern "C" int orbit2_selftest(int* result, void* stream) {
  if (!result) return O2_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(result, 0, sizeof(int), s) != hipSuccess) return O2_ERR_LAUNCH;
  bf16_t* scratch = reinterpret_cast<bf16_t*>(result + 4);  // 16-byte aligned, 512 u16
  hipLaunchKernelGGL(selftest_fill, dim3(2), dim3(256), 0, s, scratch);
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, s, result, (const bf16_t*)scratch);
  O2_CHECK_LAUNCH();
  return O2_OK;
}

// ---- memory-side latency calibration (diagnostic: tools/mall_probe.py, bench.py --mall-probe) ---------------------------------
// `blocks` workgroups of 256 threads sweep `bytes` of `buf` with 16-byte loads, `inflight` (1..8) independent loads per lane at
// a time.  Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum, a sweep of a buffer that fits the 256 MB
// Infinity Cache (but not the 32 MB of L2) gives the mean L2-miss latency of an Infinity-Cache HIT, a sweep of a multi-GB buffer
// that of an HBM read -- each at a chosen load (few blocks with one load in flight: near the unloaded latency; a full grid with
// eight: the latency under a saturating stream).  The product kernels' own mean latencies are read against these brackets.
namespace {
template <int INF>
__global__ __launch_bounds__(256) void probe_read_kernel(const u32x4* __restrict__ buf, int64_t n16, float* __restrict__ sink) {
  const int64_t stride = (int64_t)gridDim.x * 256 * INF;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * INF; i + INF <= n16; i += stride) {
    u32x4 v[INF];
#pragma unroll
    for (int j = 0; j < INF; ++j) v[j] = buf[i + j];        // plain loads: a non-temporal load would not allocate in the Infinity Cache
#pragma unroll
    for (int j = 0; j < INF; ++j) acc ^= v[j];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) sink[0] = 1.f;     // keeps the loads alive; practically never taken
}
}  // namespace

extern "C" int orbit2_probe_read(const void* buf, int64_t bytes, int blocks, int inflight, float* sink, void* stream) {
  if (!buf || !sink || bytes < 16 || blocks <= 0) return O2_ERR_ARG;
  const int64_t n16 = bytes / 16;
  hipStream_t s = (hipStream_t)stream;
  if (inflight >= 8) hipLaunchKernelGGL(probe_read_kernel<8>, dim3(blocks), dim3(256), 0, s, (const u32x4*)buf, n16, sink);
  else if (inflight >= 4) hipLaunchKernelGGL(probe_read_kernel<4>, dim3(blocks), dim3(256), 0, s, (const u32x4*)buf, n16, sink);
  else hipLaunchKernelGGL(probe_read_kernel<1>, dim3(blocks), dim3(256), 0, s, (const u32x4*)buf, n16, sink);
  O2_CHECK_LAUNCH();
  return O2_OK;
}
