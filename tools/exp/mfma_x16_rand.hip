// Developer probe: one v_mfma_f32_16x16x16_bf16 / v_mfma_f32_16x16x32_bf16 on random bf16 data vs a host reference.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// A: [16 rows][K] bf16 bits, B: [K][16 cols] bf16 bits
template <int K>
__global__ void k(const unsigned short* A, const unsigned short* B, float* out, int scale_mode) {
  const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
  f32x4 c = {0, 0, 0, 0};
  if (K == 16) {
    s16x4 a, b;
    for (int j = 0; j < 4; ++j) { a[j] = (short)A[n * 16 + 4 * g + j]; b[j] = (short)B[(4 * g + j) * 16 + n]; }
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  } else {
    s16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[n * 32 + 8 * g + j]; b[j] = (short)B[(8 * g + j) * 16 + n]; }
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + n] = c[r];
}
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
int main() {
  std::mt19937 rng(3);
  std::normal_distribution<float> nd(0.f, 1.f);
  unsigned short *dA, *dB; float* dO;
  hipMalloc(&dA, 16 * 32 * 2); hipMalloc(&dB, 32 * 16 * 2); hipMalloc(&dO, 1024);
  for (int test = 0; test < 6; ++test) {
    const int K = (test & 1) ? 32 : 16;
    const float sa = test < 2 ? 1.f : test < 4 ? 1e-3f : 1.f, sb = test < 2 ? 1.f : test < 4 ? 1e-3f : 3e-6f;   // piece-like magnitudes
    std::vector<unsigned short> A(16 * K), B(K * 16);
    for (auto& x : A) x = f2bf(sa * nd(rng));
    for (auto& x : B) x = f2bf(sb * nd(rng));
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    if (K == 16) hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 0, 0, dA, dB, dO, 0);
    else hipLaunchKernelGGL(k<32>, dim3(1), dim3(64), 0, 0, dA, dB, dO, 0);
    std::vector<float> O(256); hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0, mag = 0;
        for (int kk = 0; kk < K; ++kk) { double p = (double)bf2f(A[i * K + kk]) * bf2f(B[kk * 16 + j]); ref += p; mag += fabs(p); }
        mx = fmax(mx, fabs(O[i * 16 + j] - ref) / mag);
      }
    printf("K=%d operand scales %.0e x %.0e: max |err| / sum|ab| = %.3e\n", K, sa, sb, mx);
  }
  return 0;
}
