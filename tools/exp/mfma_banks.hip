// Developer micro-benchmark: issue interval of v_mfma_f32_16x16x4_f32 as a function of where its operands live
// (VGPR bank = register index mod 4; accumulator in VGPRs or AGPRs), one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define MF_V(c, a, b) "v_mfma_f32_16x16x4_f32 v[" c "], v" a ", v" b ", v[" c "]\n\t"
#define MF_A(c, a, b) "v_mfma_f32_16x16x4_f32 a[" c "], v" a ", v" b ", a[" c "]\n\t"

// six accumulators rotate (dependency distance 6); A in v40.., B in v50..; VARIANT picks the banks
template <int WAVES, int VARIANT>
__global__ __launch_bounds__(64 * WAVES, 1) void k(float* out, unsigned long long* cyc, int iters) {
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (VARIANT == 0)        // acc VGPR, A bank 0 (v40), B bank 1 (v49): no conflict between A and B
      asm volatile(MF_V("0:3", "40", "49") MF_V("4:7", "40", "49") MF_V("8:11", "40", "49") MF_V("12:15", "40", "49") MF_V("16:19", "40", "49") MF_V("20:23", "40", "49")
                   MF_V("0:3", "40", "49") MF_V("4:7", "40", "49") MF_V("8:11", "40", "49") MF_V("12:15", "40", "49") MF_V("16:19", "40", "49") MF_V("20:23", "40", "49")
                   ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v40","v49");
    if (VARIANT == 1)        // A and B in the same bank (v40, v48)
      asm volatile(MF_V("0:3", "40", "48") MF_V("4:7", "40", "48") MF_V("8:11", "40", "48") MF_V("12:15", "40", "48") MF_V("16:19", "40", "48") MF_V("20:23", "40", "48")
                   MF_V("0:3", "40", "48") MF_V("4:7", "40", "48") MF_V("8:11", "40", "48") MF_V("12:15", "40", "48") MF_V("16:19", "40", "48") MF_V("20:23", "40", "48")
                   ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v40","v48");
    if (VARIANT == 2)        // accumulators in AGPRs
      asm volatile(MF_A("0:3", "40", "49") MF_A("4:7", "40", "49") MF_A("8:11", "40", "49") MF_A("12:15", "40", "49") MF_A("16:19", "40", "49") MF_A("20:23", "40", "49")
                   MF_A("0:3", "40", "49") MF_A("4:7", "40", "49") MF_A("8:11", "40", "49") MF_A("12:15", "40", "49") MF_A("16:19", "40", "49") MF_A("20:23", "40", "49")
                   ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","v40","v49");
    if (VARIANT == 3)        // acc VGPR, B changes every MFMA (v49, v50, v51: banks 1, 2, 3), A changes every 3 (v40, v44)
      asm volatile(MF_V("0:3", "40", "49") MF_V("4:7", "40", "50") MF_V("8:11", "40", "51") MF_V("12:15", "44", "53") MF_V("16:19", "44", "54") MF_V("20:23", "44", "55")
                   MF_V("0:3", "40", "49") MF_V("4:7", "40", "50") MF_V("8:11", "40", "51") MF_V("12:15", "44", "53") MF_V("16:19", "44", "54") MF_V("20:23", "44", "55")
                   ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v40","v44","v49","v50","v51","v53","v54","v55");
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = 0.f;
  if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = c1 - c0; cyc[4096 + blockIdx.x * 8 + (threadIdx.x >> 6)] = r1 - r0; }
}

int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2 * 4096 * 8);
  const int iters = 20000;
#define RUN(WAVES, V, what) do { \
    hipLaunchKernelGGL((k<WAVES, V>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, 100); hipDeviceSynchronize(); \
    hipLaunchKernelGGL((k<WAVES, V>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters); hipDeviceSynchronize(); \
    std::vector<unsigned long long> h(2 * 4096); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost); \
    double s = 0, rr = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < WAVES; ++w) { s += h[b * 8 + w]; rr += h[4096 + b * 8 + w]; } \
    const double per = (WAVES / 4) * 12.0 * iters; \
    printf("%-58s %.2f memtime ticks, %.2f ns per MFMA per SIMD\n", what, s / (256 * WAVES) / per, rr / (256 * WAVES) / per * 10.0); } while (0)
  RUN(4, 0, "1 wave/SIMD, acc VGPR, A/B different banks");
  RUN(4, 1, "1 wave/SIMD, acc VGPR, A/B same bank");
  RUN(4, 2, "1 wave/SIMD, acc AGPR");
  RUN(4, 3, "1 wave/SIMD, acc VGPR, operands rotate");
  RUN(8, 0, "2 waves/SIMD, acc VGPR, A/B different banks");
  RUN(8, 1, "2 waves/SIMD, acc VGPR, A/B same bank");
  RUN(8, 2, "2 waves/SIMD, acc AGPR");
  RUN(8, 3, "2 waves/SIMD, acc VGPR, operands rotate");
  return 0;
}
