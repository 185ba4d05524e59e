// Developer micro-benchmark: the k_conv burst with 64 edges per W2 fragment (4 edge blocks) at two waves per SIMD:
// three blocks' B operand in registers (108), the fourth read from LDS one k-group ahead (ds_read_b128), full 9-fragment
// A ring, W2 streaming from L2.  Ceiling for an NB=4 kernel that fits 256 registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int LOADS, int LB>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const float* W, float* out, unsigned long long* cyc, int tiles, int n_tiles_w) {
  __shared__ __attribute__((aligned(16))) float hb[64 * 36];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 36; i += blockDim.x) hb[i] = 0.001f * i;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, n_tiles_w * 9 * 1024, 0x00020000);
  constexpr int NBR = 4 - LB;
  float Bv[NBR][36];
  for (int b = 0; b < NBR; ++b)
    for (int s = 0; s < 36; ++s) Bv[b][s] = 0.001f * (lane + 3 * b + 7 * s);
  f32x4 A[9];
  const int t0 = (blockIdx.x * 7 + wave * 131) % n_tiles_w;
  for (int s4 = 0; s4 < 9; ++s4) A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (t0 * 9 + s4) * 1024, 0));
  f32x4 tot = {0, 0, 0, 0};
  __syncthreads();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  for (int i = 0; i < tiles; ++i) {
    const int tn = (t0 + i + 1) % n_tiles_w;
    f32x4 acc[4];
    for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0, 0, 0, 0};
    f32x4 hn = *reinterpret_cast<const f32x4*>(hb + lane * 36);
#pragma unroll
    for (int s4 = 0; s4 < 9; ++s4) {
      const f32x4 av = A[s4];
      const f32x4 hc = hn;
      if (LB && s4 + 1 < 9) hn = *reinterpret_cast<const f32x4*>(hb + lane * 36 + 4 * (s4 + 1));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int b = 0; b < NBR; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Bv[b][4 * s4 + q], acc[b], 0, 0, 0);
        if (LB) acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], hc[q], acc[3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (LOADS) A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (tn * 9 + s4) * 1024, 0));
    }
    for (int b = 0; b < 4; ++b) tot += acc[b];
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
  if (lane == 0) { cyc[blockIdx.x * 8 + wave] = c1 - c0; cyc[4096 + blockIdx.x * 8 + wave] = r1 - r0; }
}

int main() {
  const int n_tiles_w = 486, tiles = 400;
  float* W; hipMalloc(&W, (size_t)n_tiles_w * 9 * 1024); hipMemset(W, 0, (size_t)n_tiles_w * 9 * 1024);
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2 * 4096 * 8);
#define RUN(WAVES, LOADS, LB, what) do { \
    hipLaunchKernelGGL((k<WAVES, LOADS, LB>), dim3(256), dim3(64 * WAVES), 0, 0, W, out, cyc, 20, n_tiles_w); hipDeviceSynchronize(); \
    hipLaunchKernelGGL((k<WAVES, LOADS, LB>), dim3(256), dim3(64 * WAVES), 0, 0, W, out, cyc, tiles, n_tiles_w); hipDeviceSynchronize(); \
    std::vector<unsigned long long> h(2 * 4096); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost); \
    double s = 0, rr = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < WAVES; ++w) { s += h[b * 8 + w]; rr += h[4096 + b * 8 + w]; } \
    const double nmf = (WAVES / 4) * 144.0 * 1.0; \
    printf("%-64s %.0f ticks, %.1f ns per tile per wave = %.2f ns per MFMA per SIMD\n", what, s / (256 * WAVES) / tiles, rr / (256 * WAVES) / tiles * 10.0, \
           rr / (256 * WAVES) / tiles * 10.0 / nmf); } while (0)
  RUN(8, 0, 1, "2 waves/SIMD, NB=4 (3 reg + 1 LDS block), no loads");
  RUN(8, 1, 1, "2 waves/SIMD, NB=4 (3 reg + 1 LDS block), streaming W2");
  RUN(4, 1, 1, "1 wave/SIMD,  NB=4 (3 reg + 1 LDS block), streaming W2");
  return 0;
}
