// Developer probe: issue interval of v_mfma_f32_16x16x32_bf16 from ONE wave with 1 / 2 / 4 independent accumulator chains,
// and from TWO waves of the same SIMD (workgroup of 8 waves) with 1 / 2 chains each.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 v = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u};
  bf16x8 a = __builtin_bit_cast(bf16x8, v), b = a;
  f32x4 c[4] = {};
  __syncthreads();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 12; ++q) c[q % CH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[q % CH], 0, 0, 0);
  }
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = c[0][0] + c[1][0] + c[2][0] + c[3][0];
  if (lane == 0) cyc[threadIdx.x >> 6] = r1 - r0;
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
  const int iters = 20000;
  for (int waves = 4; waves <= 8; waves += 4)
    for (int ch = 1; ch <= 4; ch *= 2) {
      if (ch == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
      if (ch == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
      if (ch == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
      unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      printf("%d waves/SIMD, %d chain(s) per wave: %.2f ns per MFMA per SIMD\n", waves / 4, ch, (double)h[0] * 10.0 / (12.0 * iters * (waves / 4)));
    }
  return 0;
}
