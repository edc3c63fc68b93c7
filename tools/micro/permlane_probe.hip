// prints the lane mapping of v_permlane16_swap / v_permlane32_swap on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  unsigned c = l, d = 100 + l;
  auto q = __builtin_amdgcn_permlane32_swap(c, d, false, false);
  out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int s = 0; s < 4; ++s) { printf("%s:", nm[s]); for (int i = 0; i < 64; i += 16) printf(" [%u..]", h[s * 64 + i]); printf("\n"); }
  return 0;
}
