// Developer probe (round 4): would ONE wave per SIMD with FOUR edge blocks (64 edges) beat k_conv2h's TWO waves per SIMD with two blocks each
// at the board's power limit?  Same MFMA work per CU and iteration (192 x v_mfma_f32_16x16x32_f16), the W fragments read from LDS one
// k-step ahead (ds_read_b128), NV independent v_fma_f32 per MFMA issued between the MFMAs (the contraction's stand-in):
//   A: 8 waves / CU, each fragment pair feeds  6 MFMAs (2 blocks x 3 products)   -- 64 ds_read_b128 per CU-iteration
//   B: 4 waves / CU, each fragment pair feeds 12 MFMAs (4 blocks x 3 products)   -- 32
// Round 1's finding "a wave's own vector instructions do not overlap its MFMAs" (tools/exp/mfma_shadow.hip) was taken on the fp32 matrix
// instruction, which runs at the vector rate; this asks the question again for the fp16 pipe, with random operands, over seconds.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned h) { h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15; return h; }
__device__ __forceinline__ f16x8 rnd(unsigned seed) {
  u32x4 v;
  for (int q = 0; q < 4; ++q) v[q] = (mix(seed * 4 + q + 1) & 0x87ff87ffu) | 0x38003800u;
  return __builtin_bit_cast(f16x8, v);
}

template <int NBLK, int NV2>    // NV2 = fillers per TWO MFMAs (3 = 1.5 per MFMA)
__global__ __launch_bounds__(NBLK == 2 ? 512 : 256, 1) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char frag[32 * 1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<unsigned*>(frag)[i] = (mix(i + 77) & 0x87ff87ffu) | 0x38003800u;
  __syncthreads();
  const char* fl = frag + lane * 16;
  f16x8 Bhi[NBLK][4], Blo[NBLK][4];
  for (int b = 0; b < NBLK; ++b)
    for (int s = 0; s < 4; ++s) { Bhi[b][s] = rnd(lane * 64 + b * 8 + s + blockIdx.x * 4096); Blo[b][s] = rnd(lane * 64 + b * 8 + 4 + s + blockIdx.x * 4096 + 99); }
  f32x4 accS[NBLK] = {}, accB[NBLK] = {};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 0.5f + j + lane;
  const float c = 1.0000001f;
  f16x8 nh = *reinterpret_cast<const f16x8*>(fl), nl = *reinterpret_cast<const f16x8*>(fl + 1024);
  int vi = 0;
#define FILL(n) for (int q = 0; q < (n); ++q) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[vi & 7]) : "v"(c)); ++vi; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f16x8 ah = nh, al = nl;
      nh = *reinterpret_cast<const f16x8*>(fl + ((it * 8 + 2 * s + 2) & 31) * 1024);
      nl = *reinterpret_cast<const f16x8*>(fl + ((it * 8 + 2 * s + 3) & 31) * 1024);
      __builtin_amdgcn_sched_barrier(0);
      int m = 0;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) { accS[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Blo[b][s], accS[b], 0, 0, 0); if (++m & 1) { FILL(NV2 / 2) } else { FILL(NV2 - NV2 / 2) } __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int b = 0; b < NBLK; ++b) { accS[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, Bhi[b][s], accS[b], 0, 0, 0); if (++m & 1) { FILL(NV2 / 2) } else { FILL(NV2 - NV2 / 2) } __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int b = 0; b < NBLK; ++b) { accB[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Bhi[b][s], accB[b], 0, 0, 0); if (++m & 1) { FILL(NV2 / 2) } else { FILL(NV2 - NV2 / 2) } __builtin_amdgcn_sched_barrier(0); }
    }
    if ((it & 63) == 63) for (int b = 0; b < NBLK; ++b) { accS[b] *= 1e-3f; accB[b] *= 1e-3f; }
  }
  float t = 0.f;
  for (int b = 0; b < NBLK; ++b) t += accS[b][0] + accS[b][1] + accS[b][2] + accS[b][3] + accB[b][0] + accB[b][3];
  for (int j = 0; j < 8; ++j) t += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F> static void run(const char* name, F launch, double seconds) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 10000;
  const double t0 = now();
  launch(iters); hipDeviceSynchronize();
  int n = 0; bool timing = false;
  while (true) {
    const double t = now() - t0;
    if (!timing && t > seconds / 2) { hipEventRecord(e0, 0); timing = true; n = 0; }
    if (t > seconds) break;
    for (int q = 0; q < 8; ++q) { launch(iters); ++n; }
    hipStreamSynchronize(0);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * 192.0 * (double)iters * 16384.0 * n;      // 192 MFMAs per CU-iteration in both forms
  printf("%-70s %8.1f executed TFLOP/s\n", name, flop / (ms * 1e-3) * 1e-12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
  float* out; hipMalloc(&out, 256 * 512 * 4);
#define LA(NV2) [&](int iters) { hipLaunchKernelGGL((k<2, NV2>), dim3(256), dim3(512), 0, 0, out, iters); }
#define LB(NV2) [&](int iters) { hipLaunchKernelGGL((k<4, NV2>), dim3(256), dim3(256), 0, 0, out, iters); }
  for (int rep = 0; rep < 2; ++rep) {
    run("A: 2 waves/SIMD x 2 blocks, no fillers", LA(0), seconds);
    run("B: 1 wave /SIMD x 4 blocks, no fillers", LB(0), seconds);
    run("A: 2 waves/SIMD x 2 blocks, 1.5 v_fma per MFMA", LA(3), seconds);
    run("B: 1 wave /SIMD x 4 blocks, 1.5 v_fma per MFMA", LB(3), seconds);
    run("A: 2 waves/SIMD x 2 blocks, 2 v_fma per MFMA", LA(4), seconds);
    run("B: 1 wave /SIMD x 4 blocks, 2 v_fma per MFMA", LB(4), seconds);
  }
  return 0;
}
