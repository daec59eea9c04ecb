// shader clock under load: s_memtime (clock64) against the 100 MHz constant wall clock (wall_clock64), one number per
// kernel flavour.  hipcc --offload-arch=gfx950 -O3 clk.hip -o clk && ./clk
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void busy(float* o, long long* t, int iters, int mode) {
  const long long c0 = clock64(), w0 = wall_clock64();
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a = {o[threadIdx.x], 1.f}, b = {0.5f, 0.25f}, c = {1.0001f, 0.9999f};
  float s = o[threadIdx.x + 1];
  for (int i = 0; i < iters; ++i) {
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) { a = __builtin_elementwise_fma(a, c, b); b = __builtin_elementwise_fma(b, c, a); }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) { s = fmaf(s, 1.0001f, 0.5f); }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  o[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y + b.x + s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
  float* o; long long* t;
  hipMalloc(&o, 1 << 26); hipMalloc(&t, 16);
  hipMemset(o, 0, 1 << 26);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(busy, dim3(256 * 8), dim3(256), 0, 0, o, t, 200000, mode);
      long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
      int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
      printf("mode %d (%s): s_memtime ticks %lld, wall ticks %lld (wall clock %d kHz) -> s_memtime %.1f MHz\n", mode,
             mode == 0 ? "packed fma, all CUs busy" : "scalar fma chain", h[0], h[1], wc, (double)h[0] / h[1] * wc / 1000.0);
    }
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("hipDeviceAttributeClockRate %d kHz\n", clk);
}
