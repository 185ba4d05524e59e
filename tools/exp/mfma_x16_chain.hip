// Developer probe: v_mfma_f32_16x16x16_bf16 with a non-zero accumulator input and in dependent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ inline short bf(float x) { unsigned u = __builtin_bit_cast(unsigned, x); return (short)(u >> 16); }
template <int MODE>
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  s16x4 a, b;
  for (int j = 0; j < 4; ++j) { a[j] = bf(1.f); b[j] = bf(1.f); }   // every product 1: one MFMA adds 16
  f32x4 c;
  for (int r = 0; r < 4; ++r) c[r] = 1000.f + lane * 4 + r;
  if (MODE == 0) c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  if (MODE == 1) for (int i = 0; i < 8; ++i) c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  if (MODE == 2) for (int i = 0; i < 8; ++i) { c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c)); }
  if (MODE == 3) {   // same with the 16x16x32 instruction (8 values per lane: adds 32)
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)1.f; b8[j] = (__bf16)1.f; }
    for (int i = 0; i < 8; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c, 0, 0, 0);
  }
  if (MODE == 4 || MODE == 5 || MODE == 6) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)1.f; b8[j] = (__bf16)1.f; }
    f32x4 c2 = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 4) { for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
    if (MODE == 5) { for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c2, 0, 0, 0); c += c2; }
    if (MODE == 6) { c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c2, 0, 0, 0); for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c, 0, 0, 0); c += c2; }
  }
  if (MODE == 7) {   // lane-dependent operands: a = lane group + 1 (every slot), b = 1  =>  adds 4 (1 + 2 + 3 + 4) = 40
    for (int j = 0; j < 4; ++j) a[j] = bf((float)((lane >> 4) + 1));
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 1024);
  for (int mode = 0; mode < 8; ++mode) {
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d);
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d);
    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d);
    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d);
    if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d);
    if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, d);
    if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, d);
    if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(256); hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    const float add = mode == 0 ? 16.f : mode == 3 ? 256.f : mode >= 4 && mode <= 6 ? 144.f : mode == 7 ? 40.f : 128.f;
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (h[i] != 1000.f + i + add) ++bad;
    printf("mode %d: %d of 256 outputs differ from c_in + %.0f; first: %.1f %.1f %.1f %.1f (expected %.1f ...)\n", mode, bad, add, h[0], h[1], h[2], h[3], 1000.f + add);
  }
  return 0;
}
