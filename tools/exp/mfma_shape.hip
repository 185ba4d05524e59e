// Developer micro-benchmark: which bf16 MFMA shape carries the split GEMM of k_conv2r better once the loop's other instructions
// sit between the MFMAs?  Same algorithmic work per wave either way (32 weight rows x 32 edges x K = 144 x six partial
// products), the W2 pieces from LDS (ds_read_b128, no global loads), and the same number of "other" instructions per unit of
// work, hand-interleaved one per MFMA slot as in the kernel:
//   (a) v_mfma_f32_16x16x32_bf16: two 16-row tiles x two 16-edge blocks, k padded to 160: 120 MFMAs of 16 cycles
//   (b) v_mfma_f32_32x32x16_bf16: one 32-row tile x one 32-edge block, k = 144 = 9 x 16:   54 MFMAs of 32 cycles
// fillers per 32 rows x 32 edges: 84 FMAs + 12 LDS reads (the contraction) + the fragment reads (30 / 27).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 rnd(unsigned seed) {
  u32x4 v;
  for (int q = 0; q < 4; ++q) { unsigned h = seed * 2654435761u ^ ((q + 1) * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; v[q] = (h & 0x3f803f80u) | 0x30003000u; }
  return __builtin_bit_cast(bf16x8, v);
}

template <int SHAPE, int FILL, int FEAT = 0>
__global__ __launch_bounds__(512, 1) void k(float* out, int tiles, const unsigned* src = nullptr) {
  __shared__ __attribute__((aligned(16))) char frag[32 * 1024];   // stands for the ring: fragments read lane-linear
  __shared__ __attribute__((aligned(16))) float xs[8 * 32 * 24];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 512) reinterpret_cast<unsigned*>(frag)[i] = ((i * 2654435761u) & 0x3f803f80u) | 0x30003000u;
  for (int i = threadIdx.x; i < 8 * 32 * 24; i += 512) xs[i] = 0.01f * (i % 97);
  __syncthreads();
  const float* xl = xs + (wave * 32 + (lane & 31)) * 24;
  const char* fl = frag + lane * 16;
  float z[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {1.f, 2.f, 3.f, 4.f};
  f32x4 xr = {0.f, 0.f, 0.f, 0.f};
  int fcount = 0;
  // one filler = one vector FMA, every 8th one an LDS read instead (84 + 12 per unit of work)
#define FILLER() do { if (FILL) { if ((fcount & 7) == 7) xr = *reinterpret_cast<const f32x4*>(xl + 4 * (fcount & 3)); else z[fcount & 3] = z[fcount & 3] * w[fcount & 3] + xr[fcount & 3]; ++fcount; } __builtin_amdgcn_sched_barrier(0); } while (0)
  if (SHAPE == 0) {
    bf16x8 B[2][3][5];
    for (int b = 0; b < 2; ++b) for (int i = 0; i < 3; ++i) for (int s = 0; s < 5; ++s) B[b][i][s] = rnd(lane * 31 + (b * 3 + i) * 5 + s);
    f32x4 tot = {0, 0, 0, 0};
    unsigned long long stgv[5] = {0, 0, 0, 0, 0};
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {          // two 16-row tiles = 32 rows
        f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        fcount = 0;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          if ((FEAT & 1) && (s % 2 == 0)) {            // the ring's lock-step points: k-steps 0, 2, 4
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
          }
          if (FEAT & 2) {                                // ring filling: one 8-byte load per lane now, written two k-steps later
            if (FEAT & 4) {                              // ... or a whole tile (five k-steps) later
              *reinterpret_cast<unsigned long long*>(frag + 31 * 1024 + lane * 8) = stgv[s];
              stgv[s] = *reinterpret_cast<const unsigned long long*>(src + ((t * 10 + half * 5 + s) & 1023) * 128 + lane * 2);
            } else {
              stgv[s] = *reinterpret_cast<const unsigned long long*>(src + ((t * 10 + half * 5 + s) & 1023) * 128 + lane * 2);
              *reinterpret_cast<unsigned long long*>(frag + 31 * 1024 + lane * 8) = stgv[(s + 3) % 5];
            }
          }
          const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(fl + ((half * 15 + s) & 31) * 1024);
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(fl + ((half * 15 + 5 + s) & 31) * 1024);
          const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(fl + ((half * 15 + 10 + s) & 31) * 1024);
#define M16(a, i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, B[0][i][s], acc[0], 0, 0, 0); FILLER(); acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, B[1][i][s], acc[1], 0, 0, 0); if (s < 4) FILLER(); else __builtin_amdgcn_sched_barrier(0)
          M16(a2, 0); M16(a1, 1); M16(a0, 2); M16(a1, 0); M16(a0, 1); M16(a0, 0);
#undef M16
        }
        tot += acc[0] + acc[1];
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3] + z[0] + z[1] + z[2] + z[3];
  } else {
    bf16x8 B[3][9];
    for (int i = 0; i < 3; ++i) for (int s = 0; s < 9; ++s) B[i][s] = rnd(lane * 29 + i * 9 + s);
    float tot = 0.f;
    for (int t = 0; t < tiles; ++t) {
      f32x16 acc = {};
      fcount = 0;
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(fl + (s & 31) * 1024);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(fl + ((9 + s) & 31) * 1024);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(fl + ((18 + s) & 31) * 1024);
        // 96 fillers over 54 MFMAs: two behind most of them
#define M32(a, i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B[i][s], acc, 0, 0, 0); FILLER(); if (fcount < 96 && (fcount % 9) != 0) FILLER()
        M32(a2, 0); M32(a1, 1); M32(a0, 2); M32(a1, 0); M32(a0, 1); M32(a0, 0);
#undef M32
      }
      for (int r = 0; r < 16; ++r) tot += acc[r];
    }
    out[blockIdx.x * 512 + threadIdx.x] = tot + z[0] + z[1] + z[2] + z[3];
  }
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int tiles = 2000;
  unsigned* src; hipMalloc(&src, 1024 * 128 * 4); hipMemset(src, 0x3c, 1024 * 128 * 4);
  auto report = [&](const char* shape, const char* what, float ms) {
    const double flop = 256.0 * 8 * tiles * 32.0 * 32.0 * 144.0 * 2.0;   // algorithmic fp32 flops
    printf("%-28s %-64s %8.3f ms  %7.1f fp32-equivalent TFLOP/s\n", shape, what, ms, flop / (ms * 1e-3) * 1e-12);
  };
#define RUN(KERN, shape, what) do { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0; \
    for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0, 0); hipLaunchKernelGGL((KERN), dim3(256), dim3(512), 0, 0, out, tiles, src); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } \
    report(shape, what, ms); } while (0)
  RUN((k<0, 0>), "v_mfma_f32_16x16x32_bf16", "fragments from LDS, MFMAs only");
  RUN((k<1, 0>), "v_mfma_f32_32x32x16_bf16", "fragments from LDS, MFMAs only");
  RUN((k<0, 1>), "v_mfma_f32_16x16x32_bf16", "+ interleaved fillers");
  RUN((k<1, 1>), "v_mfma_f32_32x32x16_bf16", "+ interleaved fillers");
  RUN((k<0, 1, 1>), "v_mfma_f32_16x16x32_bf16", "+ fillers + three barriers per 16-row tile");
  RUN((k<0, 1, 2>), "v_mfma_f32_16x16x32_bf16", "+ fillers + ring filling (load, ds_write two k-steps later)");
  RUN((k<0, 1, 3>), "v_mfma_f32_16x16x32_bf16", "+ fillers + barriers + ring filling");
  RUN((k<0, 1, 6>), "v_mfma_f32_16x16x32_bf16", "+ fillers + ring filling, written five k-steps after the load");
  RUN((k<0, 1, 7>), "v_mfma_f32_16x16x32_bf16", "+ fillers + barriers + ring filling written five k-steps later");
  return 0;
}
