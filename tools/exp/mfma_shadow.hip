// Developer micro-benchmark: how much independent VALU / LDS work fits in the shadow of a dense
// v_mfma_f32_16x16x4_f32 stream (a) from the SAME wave, interleaved in program order, (b) from ANOTHER wave on the
// same SIMD.  One workgroup per CU-ish (grid = 256), cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define VA(x, y) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(y))

// NV = VALU ops after every MFMA; NL: one ds_read_b128 after every NL-th MFMA (0 = none)
template <int NV, int NL>
__global__ __launch_bounds__(256, 1) void k_same(float* out, unsigned long long* cyc, int iters) {
  __shared__ f32x4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x4 acc[6];
  for (int b = 0; b < 6; ++b) acc[b] = (f32x4){0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 0.5f + j;
  f32x4 ld = {0, 0, 0, 0};
  const f32x4* lp = lds + (threadIdx.x & 63);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 36; ++m) {
      MF(acc[m % 6], a, b);
#pragma unroll
      for (int j = 0; j < NV; ++j) VA(v[j % 8], b);
      if (NL > 0 && (m % NL) == 0) {
        f32x4 t;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)(lp + ((m * 64) & 1023)) ));
        // consume 12 MFMAs later is emulated by a late add below (waitcnt placed by hand)
        asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        ld += t;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int b2 = 0; b2 < 6; ++b2) s += acc[b2][0] + acc[b2][1] + acc[b2][2] + acc[b2][3];
  for (int j = 0; j < 8; ++j) s += v[j];
  s += ld[0] + ld[1] + ld[2] + ld[3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// two waves per SIMD (512 threads): waves 0-3 run the MFMA stream, waves 4-7 run a VALU-only stream of the same
// duration budget; report both durations
template <int MODE>   // 0: MFMA waves only (others idle)  1: MFMA + VALU waves   2: VALU waves only
__global__ __launch_bounds__(512, 1) void k_two(float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[6];
  for (int b = 0; b < 6; ++b) acc[b] = (f32x4){0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f, v[8];
  for (int j = 0; j < 8; ++j) v[j] = 0.5f + j;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (MODE != 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 36; ++m) MF(acc[m % 6], a, b);
      }
  } else {
    if (MODE != 0)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 36 * 4; ++m) VA(v[m % 8], b);     // 144 VALU = 576 issue cycles per iteration
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int b2 = 0; b2 < 6; ++b2) s += acc[b2][0];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <typename F> static double run(F launch, int nwords, int pick, int iters) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  launch(out, cyc, 10); hipDeviceSynchronize();
  launch(out, cyc, iters); hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * nwords);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; int n = 0;
  for (int b = 0; b < 256; ++b) { s += h[b * nwords + pick]; ++n; }
  hipFree(out); hipFree(cyc);
  return s / n / iters;
}

int main() {
  const int iters = 2000;
#define SAME(NV, NL) printf("same wave: %d VALU/MFMA, ds_read every %d: %.1f cycles per 36 MFMA (ideal %d)\n", NV, NL, \
    run([](float* o, unsigned long long* c, int it) { hipLaunchKernelGGL((k_same<NV, NL>), dim3(256), dim3(256), 0, 0, o, c, it); }, 1, 0, iters), 36 * 32)
  SAME(0, 0); SAME(1, 0); SAME(2, 0); SAME(4, 0); SAME(6, 0); SAME(7, 0); SAME(8, 0); SAME(10, 0);
  SAME(2, 6); SAME(2, 3); SAME(4, 2); SAME(0, 1);
#define TWO(MODE, PICK, what) printf("two waves/SIMD mode %d, %s: %.1f cycles per iteration\n", MODE, what, \
    run([](float* o, unsigned long long* c, int it) { hipLaunchKernelGGL((k_two<MODE>), dim3(256), dim3(512), 0, 0, o, c, it); }, 8, PICK, iters))
  TWO(0, 0, "MFMA wave alone (36 MFMA, ideal 1152)");
  TWO(2, 4, "VALU wave alone (144 VALU, ideal 576)");
  TWO(1, 0, "MFMA wave, with VALU wave on the same SIMD");
  TWO(1, 4, "VALU wave, with MFMA wave on the same SIMD");
  return 0;
}
