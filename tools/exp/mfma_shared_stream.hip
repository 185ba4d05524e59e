// Developer micro-benchmark: every wave of a CU owns its own 16*NBK edges and walks the WHOLE W2 tile sequence; all waves
// of the CU (and of the chip) walk it from the same start, so only the leading wave misses L1 and the others should hit.
// Question: what does the burst cost per MFMA when the weight stream is shared through L1 instead of being 8 distinct
// streams per CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int NBK, int JITTER, int STREAMS = 1, int CUOFF = 1, int ASM = 0, int OVH = 0>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const float* W, float* out, unsigned long long* cyc, int tiles, int n_tiles_w) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, n_tiles_w * 9 * 1024, 0x00020000);
  float Bv[NBK][36];
  for (int b = 0; b < NBK; ++b)
    for (int s = 0; s < 36; ++s) { unsigned h = (lane * 2654435761u) ^ ((b * 36 + s + 1) * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; Bv[b][s] = (h & 1) ? 0.f : (float)(h >> 8) * (1.0f / 16777216.0f); }   // relu-like: half zeros, random mantissas
  f32x4 A[9];
  const int tbase = ((wave % STREAMS) * 131 + ((STREAMS > 1 && CUOFF) ? blockIdx.x * 17 : CUOFF == 2 ? blockIdx.x * 17 : 0)) % n_tiles_w;   // STREAMS groups of waves, each group walks from its own offset
  for (int s4 = 0; s4 < 9; ++s4) A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (tbase * 9 + s4) * 1024, 0));
  __shared__ __attribute__((aligned(16))) float xs[WAVES * 16 * NBK * 24];
  for (int i = threadIdx.x; i < WAVES * 16 * NBK * 24; i += blockDim.x) xs[i] = 0.01f * (i % 97);
  const float* xl = xs + (wave * 16 * NBK + (lane & 15)) * 24;
  float oacc[NBK][3] = {};
  f32x4 bias = {0, 0, 0, 0};
  f32x4 tot = {0, 0, 0, 0};
  __syncthreads();
  for (int k = 0; k < JITTER * ((wave * 5 + blockIdx.x * 3) % 7); ++k) __builtin_amdgcn_s_sleep(16);   // drift between waves / CUs
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < tiles; ++i) {
    const int tn = (tbase + i + 1) % n_tiles_w;
    f32x4 acc[NBK];
    for (int b = 0; b < NBK; ++b) acc[b] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 9; ++s4) {
      const f32x4 av = A[s4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
          if (ASM) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[b]) : "v"(av[q]), "v"(Bv[b][4 * s4 + q]));
          else acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Bv[b][4 * s4 + q], acc[b], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (tn * 9 + s4) * 1024, 0));
    }
    if (ASM == 2) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]));
    if (OVH) {   // k_conv-like tail of a tile: bias fetch for the next tile + per block 3 LDS reads and ~16 FMAs
      bias = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, (lane >> 4) * 16, tn * 64, 0));
#pragma unroll
      for (int b = 0; b < NBK; ++b) {
        const f32x4 v = acc[b] + bias;
        const f32x4* x4 = reinterpret_cast<const f32x4*>(xl + 16 * b * 24 + 4 * (i & 1));
        const f32x4 xa = x4[0], xb = x4[1], xc = x4[2];
        const float z0 = v[0] * xa[0] + v[1] * xa[3] + v[2] * xb[2] + v[3] * xc[1];
        const float z1 = v[0] * xa[1] + v[1] * xb[0] + v[2] * xb[3] + v[3] * xc[2];
        const float z2 = v[0] * xa[2] + v[1] * xb[1] + v[2] * xc[0] + v[3] * xc[3];
        oacc[b][0] += z1 * xa[2] - z2 * xa[1]; oacc[b][1] += z2 * xa[0] - z0 * xa[2]; oacc[b][2] += z0 * xa[1] - z1 * xa[0];
      }
    }
    for (int b = 0; b < NBK; ++b) tot += acc[b];
  }
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  for (int b = 0; b < NBK; ++b) tot[0] += oacc[b][0] + oacc[b][1] + oacc[b][2];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = r1 - r0;
}

int main() {
  const int n_tiles_w = 486, tiles = 972;
  float* W; hipMalloc(&W, (size_t)n_tiles_w * 9 * 1024);
  { std::vector<float> hw((size_t)n_tiles_w * 9 * 256); unsigned x = 12345u; for (auto& v : hw) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 26)); }   // random weights: realistic switching activity
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); }
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
#define RUN(WAVES, NBK, JIT, what) RUNS(WAVES, NBK, JIT, 1, what)
#define RUNS(WAVES, NBK, JIT, ST, what) RUNC(WAVES, NBK, JIT, ST, 1, what)
#define RUNC(WAVES, NBK, JIT, ST, CO, what) RUNA(WAVES, NBK, JIT, ST, CO, 0, what)
#define RUNA(WAVES, NBK, JIT, ST, CO, AS, what) RUNO(WAVES, NBK, JIT, ST, CO, AS, 0, what)
#define RUNO(WAVES, NBK, JIT, ST, CO, AS, OV, what) do { \
    hipLaunchKernelGGL((k<WAVES, NBK, JIT, ST, CO, AS, OV>), dim3(256), dim3(64 * WAVES), 0, 0, W, out, cyc, 20, n_tiles_w); hipDeviceSynchronize(); \
    hipLaunchKernelGGL((k<WAVES, NBK, JIT, ST, CO, AS, OV>), dim3(256), dim3(64 * WAVES), 0, 0, W, out, cyc, tiles, n_tiles_w); hipDeviceSynchronize(); \
    std::vector<unsigned long long> h(256 * 16); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost); \
    double rr = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < WAVES; ++w) rr += h[b * 16 + w]; \
    const double ns = rr / (256 * WAVES) / tiles * 10.0; \
    printf("%-66s %.1f ns per tile per wave = %.2f ns per MFMA per SIMD\n", what, ns, ns / (36.0 * NBK * (WAVES / 4))); } while (0)
  RUN(8, 3, 0, "8 waves/CU, 48 edges per wave, one shared stream, no drift");
  RUN(8, 3, 4, "8 waves/CU, 48 edges per wave, one shared stream, drifting starts");
  RUN(8, 1, 4, "8 waves/CU, 16 edges per wave, one shared stream, drifting starts");
  RUN(12, 1, 4, "12 waves/CU, 16 edges per wave, one shared stream, drifting starts");
  RUN(16, 1, 4, "16 waves/CU, 16 edges per wave, one shared stream, drifting starts");
  RUN(12, 2, 4, "12 waves/CU, 32 edges per wave, one shared stream, drifting starts");
  RUNO(12, 1, 4, 1, 2, 0, 1, "12 waves/CU, 16 edges, ONE stream + vector-path contraction and bias load per tile");
  RUNO(8, 2, 4, 1, 2, 0, 1, "8 waves/CU, 32 edges, ONE stream + vector-path contraction and bias load per tile");
  RUNO(8, 2, 4, 1, 2, 0, 0, "8 waves/CU, 32 edges, ONE stream, burst only");
  RUNO(8, 3, 4, 8, 0, 0, 1, "8 waves/CU, 48 edges, EIGHT streams + contraction (today's structure)");
  RUNA(12, 1, 4, 1, 2, 1, "12 waves/CU, 16 edges, ONE stream, inline-asm in-place MFMA");
  RUNA(12, 1, 4, 1, 2, 2, "12 waves/CU, 16 edges, ONE stream, inline-asm in-place MFMA + 18 nop states per tile");
  RUNC(8, 3, 4, 1, 2, "8 waves/CU, 48 edges, ONE stream per CU, every CU at its own position");
  RUNC(8, 3, 4, 2, 0, "8 waves/CU, 48 edges, TWO streams per CU, the same two positions on every CU");
  RUNC(8, 3, 4, 8, 0, "8 waves/CU, 48 edges, EIGHT streams per CU, the same eight positions on every CU");
  RUNC(12, 1, 4, 1, 2, "12 waves/CU, 16 edges, ONE stream per CU, every CU at its own position");
  RUNS(12, 1, 4, 2, "12 waves/CU, 16 edges per wave, TWO streams per CU");
  RUNS(12, 1, 4, 3, "12 waves/CU, 16 edges per wave, THREE streams per CU");
  RUNS(12, 1, 4, 4, "12 waves/CU, 16 edges per wave, FOUR streams per CU");
  RUNS(12, 1, 4, 12, "12 waves/CU, 16 edges per wave, TWELVE streams per CU");
  RUNS(8, 3, 4, 2, "8 waves/CU, 48 edges per wave, TWO streams per CU");
  RUNS(8, 3, 4, 4, "8 waves/CU, 48 edges per wave, FOUR streams per CU");
  RUNS(8, 3, 4, 8, "8 waves/CU, 48 edges per wave, EIGHT streams per CU (today)");
  return 0;
}
