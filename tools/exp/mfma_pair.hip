// Developer micro-benchmark: two waves per SIMD, both running the k_conv-shaped MFMA burst
// (108 MFMAs = 9 A fragments x 4 k-steps x 3 edge blocks, B operand in 108 registers), optionally with the 9+1 buffer
// loads of the next tile interleaved exactly like the kernel does.  Reports cycles per tile per wave (ideal: 6912 with
// two waves sharing a SIMD's matrix pipe, 3456 alone).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int LOADS, int SAME_TILE, int DEPTH = 1, int SYNC = 0, int AUX = 0, int WIDTH = 4, int PAUSE = 0>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const float* W, float* out, unsigned long long* cyc, int tiles, int n_tiles_w) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, n_tiles_w * 9 * 1024, 0x00020000);
  float Bv[3][36];
  for (int b = 0; b < 3; ++b)
    for (int s = 0; s < 36; ++s) Bv[b][s] = 0.001f * (lane + b + s);
  f32x4 A[9], A2[9];
  const int t0 = SYNC == 9 ? 0 : SYNC ? ((wave & 3) * (n_tiles_w / 4)) : (blockIdx.x * 7 + wave * 131) % n_tiles_w;
  for (int s4 = 0; s4 < 9; ++s4) A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (t0 * 9 + s4) * 1024, 0));
  if (DEPTH >= 2) for (int s4 = 0; s4 < 9; ++s4) A2[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (((t0 + 1) % n_tiles_w) * 9 + s4) * 1024, 0));
  f32x4 tot = {0, 0, 0, 0};
  __syncthreads();
  if (DEPTH == 3) {   // two register sets used alternately, each re-loaded in place with the tile TWO ahead (no copies)
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < tiles; i += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int tn = (t0 + i + half + 2) % n_tiles_w;
        f32x4 acc[3];
        for (int b = 0; b < 3; ++b) acc[b] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) {
          const f32x4 av = half ? A2[s4] : A[s4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Bv[b][4 * s4 + q], acc[b], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (half) A2[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (tn * 9 + s4) * 1024, 0));
          else A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (tn * 9 + s4) * 1024, 0));
        }
        for (int b = 0; b < 3; ++b) tot += acc[b];
      }
    }
    unsigned long long c1 = __builtin_readcyclecounter();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
    if (lane == 0) { cyc[blockIdx.x * 8 + wave] = c1 - c0; cyc[4096 + blockIdx.x * 8 + wave] = r1 - r0; }
    return;
  }
  if (SYNC >= 2 && wave >= 4) for (int k = 0; k < SYNC; ++k) __builtin_amdgcn_s_sleep(16);   // lag the second wave of every SIMD by SYNC x 1024 cycles
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < tiles; ++i) {
    const int tn = SAME_TILE == 1 ? t0 : SAME_TILE >= 2 ? (t0 + (i % SAME_TILE)) % n_tiles_w : (t0 + i + DEPTH) % n_tiles_w;
    f32x4 acc[3];
    for (int b = 0; b < 3; ++b) acc[b] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 9; ++s4) {
      const f32x4 av = A[s4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Bv[b][4 * s4 + q], acc[b], 0, 0, 0);
      if (PAUSE == 1) __builtin_amdgcn_s_sleep(1);
      if (PAUSE == 2) __builtin_amdgcn_s_sleep(2);
      if (PAUSE == 3) { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }
      if (LOADS && DEPTH == 1 && WIDTH == 4) A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (tn * 9 + s4) * 1024, AUX));
      if (LOADS && DEPTH == 1 && WIDTH == 1) {
        for (int c = 0; c < 4; ++c) A[s4][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW, lane * 4, (tn * 9 + s4) * 1024 + c * 256, AUX));
      }
      if (LOADS && DEPTH == 2) { A[s4] = A2[s4]; A2[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (tn * 9 + s4) * 1024, 0)); }
    }
    for (int b = 0; b < 3; ++b) tot += acc[b];
  }
  unsigned long long c1 = __builtin_readcyclecounter();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
  if (lane == 0) { cyc[blockIdx.x * 8 + wave] = c1 - c0; cyc[4096 + blockIdx.x * 8 + wave] = r1 - r0; }
}

int main() {
  const int n_tiles_w = 486, tiles = 400;
  float* W; hipMalloc(&W, (size_t)n_tiles_w * 9 * 1024); hipMemset(W, 0, (size_t)n_tiles_w * 9 * 1024);
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 2 * 512 * 4); hipMalloc(&cyc, 2 * 4096 * 8);
#define RUN(WAVES, LOADS, SAME, GRID, what) RUNX(WAVES, LOADS, SAME, 1, 0, GRID, n_tiles_w, what)
#define RUNX(WAVES, LOADS, SAME, DEPTH, SYNC, GRID, n_tiles_w, what) RUNY(WAVES, LOADS, SAME, DEPTH, SYNC, 0, 4, GRID, n_tiles_w, what)
#define RUNY(WAVES, LOADS, SAME, DEPTH, SYNC, AUX, WIDTH, GRID, n_tiles_w, what) RUNZ(WAVES, LOADS, SAME, DEPTH, SYNC, AUX, WIDTH, 0, GRID, n_tiles_w, what)
#define RUNZ(WAVES, LOADS, SAME, DEPTH, SYNC, AUX, WIDTH, PAUSE, GRID, n_tiles_w, what) do { \
    hipLaunchKernelGGL((k<WAVES, LOADS, SAME, DEPTH, SYNC, AUX, WIDTH, PAUSE>), dim3(GRID), dim3(64 * WAVES), 0, 0, W, out, cyc, 20, n_tiles_w); hipDeviceSynchronize(); \
    hipLaunchKernelGGL((k<WAVES, LOADS, SAME, DEPTH, SYNC, AUX, WIDTH, PAUSE>), dim3(GRID), dim3(64 * WAVES), 0, 0, W, out, cyc, tiles, n_tiles_w); hipDeviceSynchronize(); \
    std::vector<unsigned long long> h(2 * 4096); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost); \
    double s = 0, rr = 0; for (int b = 0; b < GRID; ++b) for (int w = 0; w < WAVES; ++w) { s += h[b * 8 + w]; rr += h[4096 + b * 8 + w]; } \
    printf("%-70s %.0f memtime ticks, %.1f ns / tile / wave\n", what, s / (GRID * WAVES) / tiles, rr / (GRID * WAVES) / tiles * 10.0); } while (0)
  RUNZ(8, 0, 0, 1, 0, 0, 4, 1, 256, 486, "2 waves/SIMD, no loads, s_sleep 1 after every 12 MFMAs");
  RUNZ(8, 0, 0, 1, 0, 0, 4, 2, 256, 486, "2 waves/SIMD, no loads, s_sleep 2 after every 12 MFMAs");
  RUNZ(8, 0, 0, 1, 0, 0, 4, 3, 256, 486, "2 waves/SIMD, no loads, 64 nop cycles after every 12 MFMAs");
  RUNZ(8, 1, 0, 1, 0, 0, 4, 1, 256, 486, "2 waves/SIMD, streaming, s_sleep 1 after every 12 MFMAs");
  RUNZ(8, 1, 0, 1, 0, 0, 4, 2, 256, 486, "2 waves/SIMD, streaming, s_sleep 2 after every 12 MFMAs");
  RUNX(8, 1, 2, 1, 9, 256, 486, "2 waves/SIMD, ALL waves cycle over the same 2 tiles (18 KB per CU: L1-resident)");
  RUNX(8, 1, 3, 1, 9, 256, 486, "2 waves/SIMD, ALL waves cycle over the same 3 tiles (27 KB per CU)");
  RUNX(8, 1, 8, 1, 9, 256, 486, "2 waves/SIMD, ALL waves cycle over the same 8 tiles (72 KB per CU)");
  RUN(8, 1, 4, 256, "2 waves/SIMD, cycling over 4 tiles per wave (288 KB per CU: L2, not L1)");
  RUN(8, 1, 16, 256, "2 waves/SIMD, cycling over 16 tiles per wave (1.1 MB per CU)");
  RUN(8, 1, 64, 256, "2 waves/SIMD, cycling over 64 tiles per wave (4.6 MB per CU)");
  RUNX(4, 1, 0, 3, 0, 256, 486, "1 wave/SIMD, streaming, two register sets (2 tiles in flight, no copies)");
  RUNX(8, 1, 0, 3, 0, 256, 486, "2 waves/SIMD, streaming, two register sets (2 tiles in flight, no copies)");
  RUNX(4, 1, 0, 2, 0, 256, 486, "1 wave/SIMD, streaming, prefetch 2 tiles ahead (with register copies)");
  RUN(4, 0, 0, 256, "1 wave/SIMD, no loads (ideal 3456)");
  RUN(4, 1, 1, 256, "1 wave/SIMD, loads of one fixed tile (L1 hits)");
  RUN(4, 1, 0, 256, "1 wave/SIMD, streaming W2 from L2");
  RUN(8, 0, 0, 256, "2 waves/SIMD, no loads (ideal 6912)");
  RUN(8, 1, 1, 256, "2 waves/SIMD, loads of one fixed tile (L1 hits)");
  RUN(8, 1, 0, 256, "2 waves/SIMD, streaming W2 from L2 (one workgroup per CU)");
  RUNX(8, 1, 0, 1, 0, 256, 243, "2 waves/SIMD, streaming a 2.2 MB W2 (fits one XCD's L2)");
  RUNX(8, 1, 0, 1, 0, 256, 121, "2 waves/SIMD, streaming a 1.1 MB W2");
  RUNX(8, 1, 0, 2, 0, 256, 486, "2 waves/SIMD, 4.5 MB, prefetch 2 tiles ahead");
  RUNX(8, 1, 0, 1, 1, 256, 486, "2 waves/SIMD, 4.5 MB, all workgroups walk the same tiles in step");
  RUNX(8, 1, 0, 2, 1, 256, 486, "2 waves/SIMD, 4.5 MB, in step + prefetch 2 ahead");
  RUNX(8, 1, 0, 1, 0, 64, 486, "2 waves/SIMD, 4.5 MB, only 64 workgroups (8 per XCD)");
  RUNX(8, 1, 0, 1, 2, 256, 486, "in step, second wave lags 2k cycles");
  RUNX(8, 1, 0, 1, 3, 256, 486, "in step, second wave lags 3k cycles");
  RUNX(8, 1, 0, 1, 4, 256, 486, "in step, second wave lags 4k cycles");
  RUNX(8, 1, 0, 1, 6, 256, 486, "in step, second wave lags 6k cycles");
  RUNX(8, 1, 0, 1, 1, 256, 486, "in step, no lag");
  return 0;
  RUNY(8, 1, 0, 1, 0, 1, 4, 256, 486, "aux=1 (sc0)");
  RUNY(8, 1, 0, 1, 0, 2, 4, 256, 486, "aux=2 (sc1/slc)");
  RUNY(8, 1, 0, 1, 0, 3, 4, 256, 486, "aux=3");
  RUNY(8, 1, 0, 1, 0, 16, 4, 256, 486, "aux=16");
  RUNY(8, 1, 0, 1, 0, 17, 4, 256, 486, "aux=17");
  RUNY(8, 1, 0, 1, 0, 0, 1, 256, 486, "dword loads (4 per fragment, 256 B per wave each)");
  return 0;
}
