// Developer micro-benchmark (round 6): issue cost of the vector instructions of k_convz's cut (fp32 -> two fp16 pieces) from ONE wave per SIMD and from two,
// independent and as a dependent chain: cycles per wave-instruction from s_memtime around an unrolled stream of 512.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/valu_rate tools/exp/valu_rate.hip && gpurun -- tools/exp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void k(float* out, long long* cyc, float a, float b) {
  float v0 = a + threadIdx.x, v1 = b, v2 = a * 2, v3 = b * 3, v4 = a - 1, v5 = b + 5, v6 = a * 7, v7 = b - 3;
  unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0;
  __shared__ f32x4 sm[1024];
  f32x4 q = {v0, v1, v2, v3};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    if (OP == 0) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(b));) }
    if (OP == 1) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(b));) }
    if (OP == 2) { REP64(asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %5, %6\n v_cvt_pk_f16_f32 %2, %6, %7\n v_cvt_pk_f16_f32 %3, %7, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));) }
    if (OP == 3) { REP64(asm volatile("v_fma_mix_f32 %0, %4, %5, -%8\n v_fma_mix_f32 %1, %5, %6, -%8\n v_fma_mix_f32 %2, %6, %7, -%8\n v_fma_mix_f32 %3, %7, %4, -%8" : "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(u0));) }
    if (OP == 4) {   // the cut of one pair, as hipcc emits it: 2 mul, cvt_pk, 2 fma_mix, cvt_pk (dependent)
      REP64(asm volatile("v_mul_f32 %0, %4, %6\n v_mul_f32 %1, %5, %6\n v_cvt_pk_f16_f32 %2, %0, %1\n v_fma_mix_f32 %0, %4, %6, -%2\n v_fma_mix_f32 %1, %5, %6, -%2 op_sel:[0,0,1]\n v_cvt_pk_f16_f32 %3, %0, %1" : "+v"(v4), "+v"(v5), "+v"(u0), "+v"(u1) : "v"(v0), "v"(v1), "v"(b));) }
    if (OP == 5) {   // two cuts interleaved
      REP64(asm volatile("v_mul_f32 %0, %8, %12\n v_mul_f32 %4, %10, %12\n v_mul_f32 %1, %9, %12\n v_mul_f32 %5, %11, %12\n v_cvt_pk_f16_f32 %2, %0, %1\n v_cvt_pk_f16_f32 %6, %4, %5\n v_fma_mix_f32 %0, %8, %12, -%2\n v_fma_mix_f32 %4, %10, %12, -%6\n v_fma_mix_f32 %1, %9, %12, -%2\n v_fma_mix_f32 %5, %11, %12, -%6\n v_cvt_pk_f16_f32 %3, %0, %1\n v_cvt_pk_f16_f32 %7, %4, %5"
                         : "+v"(v4), "+v"(v5), "+v"(u0), "+v"(u1), "+v"(v6), "+v"(v7), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(b));) }
    if (OP == 6) { REP64(asm volatile("v_fma_mixlo_f16 %0, %4, %6, 0\n v_fma_mixhi_f16 %0, %5, %6, 0\n v_fma_mixlo_f16 %1, %4, %6, -%0 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %1, %5, %6, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(b));) }
    if (OP == 7) { REP64(asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:8192\n ds_write_b128 %0, %1 offset:12288" :: "v"((unsigned)(threadIdx.x & 63) * 16), "v"(q) : "memory");) }
    if (OP == 8) { REP64(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %1, a0\n v_accvgpr_write_b32 a1, %2\n v_accvgpr_read_b32 %3, a1" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) :: "a0", "a1");) }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0 && blockIdx.x == 0) cyc[threadIdx.x / 64] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(u0 ^ u1 ^ u2 ^ u3) + sm[threadIdx.x & 1023][0];
}
template <int OP>
void run(const char* name, int per_rep) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.9999f);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-44s %d waves/SIMD: %.2f cycles per instruction (wave 0)\n", name, threads / 256, (double)h[0] / (8.0 * 64 * per_rep));
  }
}
int main() {
  run<0>("v_mul_f32 independent", 4);
  run<1>("v_mul_f32 dependent chain", 4);
  run<2>("v_cvt_pk_f16_f32 independent", 4);
  run<3>("v_fma_mix_f32 independent", 4);
  run<4>("cut of one pair (6 instrs, dependent)", 6);
  run<5>("two cuts interleaved (12 instrs)", 12);
  run<6>("cut with v_fma_mixlo/hi_f16 (4 instrs)", 4);
  run<7>("ds_write_b128", 4);
  run<8>("v_accvgpr_write + read", 4);
  return 0;
}
