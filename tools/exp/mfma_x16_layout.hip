// Developer probe: which (lane group, slot) of the A operand of v_mfma_f32_16x16x16_bf16 meets which (lane group, slot) of B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ inline short bf(float x) { unsigned u = __builtin_bit_cast(unsigned, x); return (short)(u >> 16); }
__global__ void k(float* out) {
  const int lane = threadIdx.x, la = blockIdx.x >> 2, ja = blockIdx.x & 3;
  s16x4 a = {0, 0, 0, 0}, b;
  if (lane == la) a[ja] = bf(1.f);
  for (int j = 0; j < 4; ++j) b[j] = bf((float)(lane * 4 + j + 1));   // 1..256: exact in bf16
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[((size_t)blockIdx.x * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 256 * 4);
  hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, d);
  std::vector<float> h(256 * 256); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  for (int ga = 0; ga < 4; ++ga)
    for (int ja = 0; ja < 4; ++ja) {
      const int la = ga * 16 + 5, blk = la * 4 + ja;
      int row = -1, lb = -1, jb = -1, nz = 0;
      for (int i = 0; i < 16; ++i)
        for (int c = 0; c < 16; ++c) { float v = h[(blk * 16 + i) * 16 + c]; if (v != 0.f) { ++nz; if (c == 3) { row = i; lb = ((int)v - 1) / 4; jb = ((int)v - 1) % 4; } } }
      printf("A lane group %d slot %d (lane %d): nonzeros %d, row %d; at col 3 meets B lane %d (group %d, col %d) slot %d\n", ga, ja, la, nz, row, lb, lb >> 4, lb & 15, jb);
    }
  return 0;
}
