// What the fp16 matrix pipe of THIS board sustains: a bare stream of v_mfma_f32_16x16x32_f16 with random operands on every CU, two waves
// per SIMD, nothing else -- the measurement of tools/exp/mfma_power.hip (round 3) as a library entry so that bench.py can take it on the
// board it runs on instead of quoting a constant from another one.  The firmware holds such a stream at the board's
// power cap by lowering the clock (round 3's board: 1.94 PFLOP/s at 2.05 GHz; the nominal 2.5 PFLOP/s is reached with all-zero operands
// only), so the number differs from board to board like the conv kernel's does -- dbfr_probe_mfma_f16 is the denominator of
// roofline.frac_of_sustained.
#include <chrono>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned probe_mix(unsigned h) { h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15; return h; }
__device__ __forceinline__ f16x8 probe_rnd(unsigned seed) {          // +-[0.5, 2) with random significands
  u32x4 v;
  for (int q = 0; q < 4; ++q) v[q] = (probe_mix(seed * 4 + q + 1) & 0x87ff87ffu) | 0x38003800u;
  return __builtin_bit_cast(f16x8, v);
}

// 24 MFMAs per iteration in k_conv2h's order within a k-step (hi_w lo_h0, hi_w lo_h1, lo_w hi_h0, lo_w hi_h1, hi_w hi_h0, hi_w hi_h1)
__global__ __launch_bounds__(512, 1) void k_probe_mfma(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 A[8], B[8];
  for (int j = 0; j < 8; ++j) { A[j] = probe_rnd(lane * 16 + j); B[j] = probe_rnd(lane * 16 + 8 + j + blockIdx.x * 1024); }
  f32x4 acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f16x8 ah = A[j], al = A[j + 1];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[j + 1], acc[0], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[(j + 5) & 7], acc[2], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, B[j], acc[0], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, B[(j + 4) & 7], acc[2], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[j], acc[1], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[(j + 4) & 7], acc[3], 0, 0, 0);
    }
    if ((it & 63) == 63) for (int q = 0; q < 4; ++q) acc[q] *= 1e-3f;     // (keeps the accumulators finite)
  }
  const f32x4 t = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 512 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

extern "C" int dbfr_probe_mfma_f16(double seconds, double* tflops, void* hip_stream) {
  if (!tflops || !(seconds > 0.0) || seconds > 60.0) { dbfr_set_error("dbfr_probe_mfma_f16: bad argument"); return DBFR_ERR_ARG; }
  hipStream_t st = (hipStream_t)hip_stream;
  const int n_cu = dbfr_current_cu_count();
  float* out = nullptr;
  HIPCHECK(hipMalloc(&out, (size_t)n_cu * 512 * sizeof(float)));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
  const int iters = 20000;                                   // ~10 ms per launch
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  long n = 0;
  bool timing = false;
  int rc = DBFR_OK;
  while (rc == DBFR_OK) {          // the first half settles the clock, the rate is taken over the second half
    const double t = now() - t0;
    if (!timing && t > seconds / 2) { if (hipEventRecord(e0, st) != hipSuccess) rc = DBFR_ERR_HIP; timing = true; n = 0; }
    if (t > seconds) break;
    for (int q = 0; q < 8; ++q) { hipLaunchKernelGGL(k_probe_mfma, dim3(n_cu), dim3(512), 0, st, out, iters); ++n; }
    if (hipStreamSynchronize(st) != hipSuccess) rc = DBFR_ERR_HIP;
  }
  float ms = 0.f;
  if (rc == DBFR_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = DBFR_ERR_HIP;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
  if (rc != DBFR_OK || ms <= 0.f || n <= 0) { dbfr_set_error("dbfr_probe_mfma_f16: HIP error"); return DBFR_ERR_HIP; }
  // per launch: n_cu workgroups x 8 waves x iters x 24 MFMAs x 16*16*32*2 flop
  *tflops = (double)n_cu * 8.0 * iters * 24.0 * 16384.0 * (double)n / (ms * 1e-3) * 1e-12;
  return DBFR_OK;
}
