// Developer micro-benchmark: same burst as mfma_pair.hip, but the W2 fragments of the next tile travel L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR return path) into a per-wave ring and are read back with ds_read_b128.
// Question: is the ~9.6 B/clk/CU fill limit a property of the L2->L1 path or of the VGPR return path?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int MODE>   // MODE 0: no loads, 1: LDS-DMA stream, 2: LDS-DMA of one fixed tile
__global__ __launch_bounds__(64 * WAVES, 1) void k(const float* W, float* out, unsigned long long* cyc, int tiles, int n_tiles_w) {
  __shared__ __attribute__((aligned(16))) float ring[WAVES][9][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float Bv[3][36];
  for (int b = 0; b < 3; ++b)
    for (int s = 0; s < 36; ++s) Bv[b][s] = 0.001f * (lane + b + s);
  const int t0 = (blockIdx.x * 7 + wave * 131) % n_tiles_w;
  for (int s4 = 0; s4 < 9; ++s4)
    for (int q = 0; q < 4; ++q) ring[wave][s4][lane * 4 + q] = W[((size_t)t0 * 9 + s4) * 256 + lane * 4 + q];
  f32x4 tot = {0, 0, 0, 0};
  __syncthreads();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < tiles; ++i) {
    const int tn = MODE == 2 ? t0 : (t0 + i + 1) % n_tiles_w;
    f32x4 acc[3];
    for (int b = 0; b < 3; ++b) acc[b] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 9; ++s4) {
      // fragment s4 of this tile was requested 9 DMA instructions ago: at most 8 younger ones may still be in flight
      if (MODE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      f32x4 av;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(av) : "v"((unsigned)(size_t)(&ring[wave][s4][lane * 4])) : "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Bv[b][4 * s4 + q], acc[b], 0, 0, 0);
      if (MODE) {
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(W + ((size_t)tn * 9 + s4) * 256 + lane * 4),
                                         (void __attribute__((address_space(3)))*)(&ring[wave][s4][0]), 16, 0, 0);
      }
    }
    for (int b = 0; b < 3; ++b) tot += acc[b];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long c1 = __builtin_readcyclecounter();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
  if (lane == 0) { cyc[blockIdx.x * 8 + wave] = c1 - c0; cyc[4096 + blockIdx.x * 8 + wave] = r1 - r0; }
}

int main() {
  const int n_tiles_w = 486, tiles = 400;
  float* W; hipMalloc(&W, (size_t)n_tiles_w * 9 * 1024); hipMemset(W, 0, (size_t)n_tiles_w * 9 * 1024);
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 2 * 512 * 4); hipMalloc(&cyc, 2 * 4096 * 8);
#define RUN(WAVES, MODE, what) do { \
    hipLaunchKernelGGL((k<WAVES, MODE>), dim3(256), dim3(64 * WAVES), 0, 0, W, out, cyc, 20, n_tiles_w); hipDeviceSynchronize(); \
    hipLaunchKernelGGL((k<WAVES, MODE>), dim3(256), dim3(64 * WAVES), 0, 0, W, out, cyc, tiles, n_tiles_w); hipDeviceSynchronize(); \
    std::vector<unsigned long long> h(2 * 4096); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost); \
    double s = 0, rr = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < WAVES; ++w) { s += h[b * 8 + w]; rr += h[4096 + b * 8 + w]; } \
    printf("%-58s %.0f memtime ticks / tile / wave, %.1f ns (100 MHz realtime) => %.3f MFMA-cycles per ns if the pipe were full\n", what, s / (256 * WAVES) / tiles, \
           rr / (256 * WAVES) / tiles * 10.0, (WAVES == 8 ? 6912.0 : 3456.0) / (rr / (256 * WAVES) / tiles * 10.0)); } while (0)
  RUN(4, 0, "1 wave/SIMD, A from LDS, no loads (ideal 3456)");
  RUN(8, 0, "2 waves/SIMD, A from LDS, no loads (ideal 6912)");
  RUN(8, 2, "2 waves/SIMD, LDS-DMA of one fixed tile");
  RUN(8, 1, "2 waves/SIMD, LDS-DMA stream of W2 from L2");
  RUN(4, 1, "1 wave/SIMD, LDS-DMA stream of W2 from L2 (ideal 3456)");
  return 0;
}
