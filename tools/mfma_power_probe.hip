// What the chip sustains on bf16 MFMAs alone, by instruction shape: 32x32x16 (16 accumulator registers, operands reused
// across 32 rows/columns) against 16x16x32 (4 accumulator registers, half the reuse) -- same FLOPs per instruction byte,
// random operands, one wave per SIMD on every CU, ~1 s per case so the power management has settled.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// FILL: per 32768 MFMA FLOPs one v_exp_f32, three v_fma_f32 and one ds_read_b128 (about the vector / LDS density of the attention
// forward) issued between the MFMAs of either shape
#define O2_FILL()                                                                                     \
  if (FILL) {                                                                                         \
    e0 = __builtin_amdgcn_exp2f(e0 * 0.999f);                                                         \
    f0 = __builtin_fmaf(f0, 0.99f, e0); f1 = __builtin_fmaf(f1, 0.98f, f0); f2 = __builtin_fmaf(f2, 0.97f, f1); \
    const uint4 t = lds[(threadIdx.x + (it & 7) * 64) & 1023];                                        \
    asm volatile("" : "+v"(e0), "+v"(f2) : "v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));                  \
  }
template <int SHAPE, int NOPS, bool FILL>
__global__ __launch_bounds__(256, 1) void probe(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x + blockIdx.x * 256;
  __shared__ uint4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = src[lane + i];
  __syncthreads();
  float e0 = 0.5f, f0 = 0.1f, f1 = 0.2f, f2 = 0.3f;
  bf16x8 a[NOPS], b[NOPS];
  for (int i = 0; i < NOPS; ++i) {
    uint4 x = src[(lane * NOPS + i) * 2], y = src[(lane * NOPS + i) * 2 + 1];
    a[i] = *reinterpret_cast<bf16x8*>(&x);
    b[i] = *reinterpret_cast<bf16x8*>(&y);
  }
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) c[i][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + j) % NOPS], c[j], 0, 0, 0); O2_FILL(); }
    }
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += c[i][k];
  } else {
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) for (int k = 0; k < 4; ++k) c[i][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(i + j) % NOPS], c[j], 0, 0, 0); if (j & 1) { O2_FILL(); } }
    }
    for (int i = 0; i < 8; ++i) for (int k = 0; k < 4; ++k) s += c[i][k];
  }
  out[lane] = s + e0 + f2;
}

int main() {
  const int NOPS = 8, blocks = 256 * 4, threads = 256;
  const size_t n = (size_t)blocks * threads * NOPS * 2;
  std::vector<uint4> h(n);
  srand(1);
  for (auto& v : h) {   // random bf16 in about [-2, 2): random sign, exponent 120..127, random mantissa
    unsigned w[4];
    for (int k = 0; k < 4; ++k) {
      unsigned lo = ((rand() & 1) << 15) | ((120 + (rand() & 7)) << 7) | (rand() & 127);
      unsigned hi = ((rand() & 1) << 15) | ((120 + (rand() & 7)) << 7) | (rand() & 127);
      w[k] = lo | (hi << 16);
    }
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  uint4* d; float* o;
  hipMalloc(&d, n * sizeof(uint4)); hipMalloc(&o, (size_t)blocks * threads * 4);
  hipMemcpy(d, h.data(), n * sizeof(uint4), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int fill = 0; fill < 2; ++fill)
  for (int rep = 0; rep < 2; ++rep)
    for (int shape : {32, 16, 32, 16}) {
      const int iters = 200000;
      const double flops = shape == 32 ? 2.0 * 32 * 32 * 16 * NOPS * 4 : 2.0 * 16 * 16 * 32 * NOPS * 8;
      float ms = 0;
      for (int pass = 0; pass < 3; ++pass) {
        hipEventRecord(e0);
        if (shape == 32 && !fill) hipLaunchKernelGGL((probe<32, NOPS, false>), dim3(blocks), dim3(threads), 0, 0, d, o, iters);
        else if (shape == 32) hipLaunchKernelGGL((probe<32, NOPS, true>), dim3(blocks), dim3(threads), 0, 0, d, o, iters);
        else if (!fill) hipLaunchKernelGGL((probe<16, NOPS, false>), dim3(blocks), dim3(threads), 0, 0, d, o, iters);
        else hipLaunchKernelGGL((probe<16, NOPS, true>), dim3(blocks), dim3(threads), 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double tf = flops * iters * (blocks * threads / 64) / (ms * 1e-3) / 1e12;
      printf("%s mfma %2dx%2d: %8.2f ms  %7.1f TFLOP/s  (%.3f of 2.5 PF)\n", fill ? "exp + 3 fma + ds_read_b128 per 32768 FLOP," : "bare loop,", shape, shape, ms, tf, tf / 2500.0);
    }
  return 0;
}
