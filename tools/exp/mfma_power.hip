// Developer probe (round 3): what the fp16 matrix pipe sustains at the board's power limit, by instruction shape, with random
// operands (all-zero operands draw far less), and what the W2 fragment reads from LDS add.  Every variant does the same work per
// wave and iteration (24 x 16x16x32 = 12 x 32x32x16 MFMAs = 393 216 flop per lane group) and the same LDS bytes per flop
// (8 ds_read_b128 per iteration, k_conv2h's ratio: one hi + one lo fragment per six 16x16x32 MFMAs).  Each variant runs ~8 s
// back to back; the rate is taken over the second half, when the firmware has settled the clock.  Sample
// `rocm-smi --showpower --showclocks` next to it (tools/exp/mfma_power.sh) for the watts and the clock.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned h) { h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15; return h; }
__device__ __forceinline__ f16x8 rnd(unsigned seed, bool zero) {
  u32x4 v;
  for (int q = 0; q < 4; ++q) v[q] = zero ? 0u : ((mix(seed * 4 + q + 1) & 0x87ff87ffu) | 0x38003800u);   // +-[0.5, 2), random significand
  return __builtin_bit_cast(f16x8, v);
}

// ORDER (SHAPE 0 only): 0 = k_conv2h's order within a k-step (hi_w lo_h0, hi_w lo_h1, lo_w hi_h0, lo_w hi_h1, hi_w hi_h0, hi_w hi_h1: three
// of six consecutive MFMAs share an operand register with their predecessor), 1 = every consecutive pair shares one (hi_w lo_h0, hi_w lo_h1,
// hi_w hi_h1, lo_w hi_h1, lo_w hi_h0, hi_w hi_h0) -- does operand reuse between consecutive MFMAs lower the power?
template <int SHAPE, int LDSRD, int ORDER = 0>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int zero) {
  __shared__ __attribute__((aligned(16))) char frag[32 * 1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 512) reinterpret_cast<unsigned*>(frag)[i] = zero ? 0u : ((mix(i + 77) & 0x87ff87ffu) | 0x38003800u);
  __syncthreads();
  const char* fl = frag + lane * 16;
  f16x8 A[8], B[8];
  for (int j = 0; j < 8; ++j) { A[j] = rnd(lane * 16 + j, zero); B[j] = rnd(lane * 16 + 8 + j + blockIdx.x * 1024, zero); }
  if (SHAPE == 0) {
    f32x4 acc[4] = {};
    f16x8 nh = *reinterpret_cast<const f16x8*>(fl), nl = *reinterpret_cast<const f16x8*>(fl + 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        f16x8 ah = A[j], al = A[j + 1];
        if (LDSRD == 1) {
          ah = *reinterpret_cast<const f16x8*>(fl + ((it * 8 + j) & 31) * 1024);
          al = *reinterpret_cast<const f16x8*>(fl + ((it * 8 + j + 1) & 31) * 1024);
        }
        if (LDSRD == 2) {                            // fragments read one k-step ahead, as k_conv2h does
          ah = nh; al = nl;
          nh = *reinterpret_cast<const f16x8*>(fl + ((it * 8 + j + 2) & 31) * 1024);
          nl = *reinterpret_cast<const f16x8*>(fl + ((it * 8 + j + 3) & 31) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
        // two edge blocks x (hi*lo, lo*hi -> small accumulator; hi*hi -> large accumulator)
        if (ORDER == 0) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[j + 1], acc[0], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[(j + 5) & 7], acc[2], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, B[j], acc[0], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, B[(j + 4) & 7], acc[2], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[j], acc[1], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[(j + 4) & 7], acc[3], 0, 0, 0);
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[j + 1], acc[0], 0, 0, 0);            // hi_w lo_h0
          __builtin_amdgcn_sched_barrier(0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[(j + 5) & 7], acc[2], 0, 0, 0);      // hi_w lo_h1   (A kept)
          __builtin_amdgcn_sched_barrier(0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[(j + 4) & 7], acc[3], 0, 0, 0);      // hi_w hi_h1   (A kept)
          __builtin_amdgcn_sched_barrier(0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, B[(j + 4) & 7], acc[2], 0, 0, 0);      // lo_w hi_h1   (B kept)
          __builtin_amdgcn_sched_barrier(0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, B[j], acc[0], 0, 0, 0);                // lo_w hi_h0   (A kept)
          __builtin_amdgcn_sched_barrier(0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[j], acc[1], 0, 0, 0);                // hi_w hi_h0   (B kept)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if ((it & 63) == 63) for (int q = 0; q < 4; ++q) acc[q] *= 1e-3f;
    }
    f32x4 t = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 512 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
  } else {
    f32x16 acc[2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        f16x8 ah = A[j], al = A[j + 1];
        if (LDSRD) {
          ah = *reinterpret_cast<const f16x8*>(fl + ((it * 8 + j) & 31) * 1024);
          al = *reinterpret_cast<const f16x8*>(fl + ((it * 8 + j + 1) & 31) * 1024);
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B[j + 1], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, B[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B[j], acc[1], 0, 0, 0);
      }
      if ((it & 63) == 63) for (int q = 0; q < 2; ++q) acc[q] *= 1e-3f;
    }
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += acc[0][r] + acc[1][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
  }
}

static double now() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }

template <typename F> static void run(const char* name, F launch, double seconds) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;                          // ~10 ms per launch
  const double t0 = now();
  launch(iters); hipDeviceSynchronize();
  int n = 0; bool timing = false; double tstart = 0;
  while (true) {
    const double t = now() - t0;
    if (!timing && t > seconds / 2) { hipEventRecord(e0, 0); timing = true; n = 0; tstart = now(); }
    if (t > seconds) break;
    for (int q = 0; q < 8; ++q) { launch(iters); ++n; }
    hipStreamSynchronize(0);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * 8 * (double)iters * 24.0 * 16384.0 * n;
  printf("%-58s wall %.1f .. %.1f  measured %.1f .. %.1f : %8.1f executed TFLOP/s\n", name, t0, now(), tstart, now(), flop / (ms * 1e-3) * 1e-12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 8.0;
  float* out; hipMalloc(&out, 256 * 512 * 4);
#define L(S, R, Z) [&](int iters) { hipLaunchKernelGGL((k<S, R>), dim3(256), dim3(512), 0, 0, out, iters, Z); }
#define LO(R, O) [&](int iters) { hipLaunchKernelGGL((k<0, R, O>), dim3(256), dim3(512), 0, 0, out, iters, 0); }
  if (argc > 2) {   // operand-reuse order probe
    for (int rep = 0; rep < 2; ++rep) {
      run("order 0 (kernel's), registers", LO(0, 0), seconds);
      run("order 1 (every pair shares an operand), registers", LO(0, 1), seconds);
      run("order 0 (kernel's), W fragments from LDS one k-step ahead", LO(2, 0), seconds);
      run("order 1 (every pair shares an operand), LDS one k-step ahead", LO(2, 1), seconds);
    }
    return 0;
  }
  run("16x16x32 f16, operands in registers, random data", L(0, 0, 0), seconds);
  run("32x32x16 f16, operands in registers, random data", L(1, 0, 0), seconds);
  run("16x16x32 f16, W fragments from LDS, random data", L(0, 1, 0), seconds);
  run("32x32x16 f16, W fragments from LDS, random data", L(1, 1, 0), seconds);
  run("16x16x32 f16, W fragments from LDS one k-step ahead", L(0, 2, 0), seconds);
  run("16x16x32 f16, operands in registers, ALL-ZERO data", L(0, 0, 1), seconds);
  run("32x32x16 f16, operands in registers, ALL-ZERO data", L(1, 0, 1), seconds);
  return 0;
}
