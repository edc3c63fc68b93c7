// Microbenchmark: LDS read throughput per wave-instruction, ds_read_b128 vs ds_read_b64_tr_b16, using the exact
// address patterns of the GEMM kernels.  Prints cycles per instruction per wave (one wave per SIMD, 4 waves/CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((int*)smem)[i] = i;
  __syncthreads();
  int acc = 0;
  // MODE 0: b128 row read, 128-B rows, swizzled chunk (GEMM128 KC);  MODE 1: tr read, 256-B rows, swizzled granule
  // MODE 2: b128 unswizzled (conflicting);  MODE 3: tr read unswizzled
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r0 = ((u + it) & 7) * 16;
      if (MODE == 0 || MODE == 2) {
        const int row = r0 + (lane & 15), chunk = (u & 1) * 4 + (lane >> 4);
        const int c = MODE == 0 ? (chunk ^ (row & 7)) : chunk;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * 128 + (c << 4));
        acc += v[0] + v[7];
      } else {
        const int krow = (u & 1) * 32 + 8 * g + q;
        const int sw = MODE == 1 ? (q | ((g & 1) << 2)) : 0;
        const char* a = smem + krow * 256 + ((((r0 >> 4) ^ sw)) << 5) + 8 * pp;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4, a));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4, a + 4 * 256));
        acc += lo[0] + hi[3];
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 0x7fffffff) out[1] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  const int iters = 2000;
  const char* names[4] = {"ds_read_b128 swizzled (16 B/lane)", "2x ds_read_b64_tr_b16 swizzled (2x8 B/lane)",
                          "ds_read_b128 unswizzled", "2x ds_read_b64_tr_b16 unswizzled"};
  for (int m = 0; m < 4; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, iters);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, iters);
      hipDeviceSynchronize();
    }
    unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-48s %.2f cycles per 16-byte-per-lane fragment (per wave, 4 waves/CU)\n", names[m], (double)h / (iters * 8.0));
  }
  return 0;
}
