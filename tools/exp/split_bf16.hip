// Developer micro-benchmark: the radial-MLP GEMM with fp32 operands split into bf16 pieces (a = a1 + a2 + a3, 8 significand
// bits each => exact) and the partial products a_i b_j evaluated on the bf16 matrix pipe with fp32 accumulation.
//   part A  accuracy of w[row][edge] = sum_k W[row][k] h[k][edge], K = 144, against an fp64 reference:
//           native v_mfma_f32_16x16x4_f32 chain vs 3 / 6 / 9 partial products on v_mfma_f32_16x16x32_bf16 (K padded to 160)
//   part B  throughput of the k_conv2-shaped sweep (edge-owner waves, h pieces resident in registers, W tiles shared by the
//           workgroup through LDS-DMA double buffering, or fetched per wave through L1), with a k_conv-like contraction tail
//   part C  issue interval of v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_16x16x16_bf16 from one wave
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned short bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__host__ __device__ inline float bf16_f(unsigned short b) {
  unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
__host__ __device__ inline void split3(float x, unsigned short* p) {
  p[0] = bf16_rne(x);
  float r = x - bf16_f(p[0]);
  p[1] = bf16_rne(r);
  r -= bf16_f(p[1]);
  p[2] = bf16_rne(r);
}

// K permutation shared by both operands: step s (32 k's), lane group g, slot j = 4 mm + r  <->  k = 16 (2 s + mm) + 4 g + r
__host__ __device__ inline int kperm(int s, int g, int j) { return 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3); }

// ------------------------------------------------------------------ part A
// W: [tiles][16][144] fp32, H: [tiles][144][16] fp32; out: [tiles][4 variants][16][16]
__global__ void k_acc(const float* W, const float* H, float* out) {
  const int lane = threadIdx.x, n = lane & 15, g = lane >> 4, t = blockIdx.x;
  const float* w = W + (size_t)t * 16 * 144;
  const float* h = H + (size_t)t * 144 * 16;
  f32x4 acc = {0, 0, 0, 0};
  for (int k = 0; k < 144; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n * 144 + k + g], h[(k + g) * 16 + n], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(((size_t)t * 4 + 0) * 16 + 4 * g + r) * 16 + n] = acc[r];
  bf16x8 A[3][5], B[3][5];
  for (int s = 0; s < 5; ++s)
    for (int j = 0; j < 8; ++j) {
      const int k = kperm(s, g, j);
      unsigned short pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
      if (k < 144) { split3(w[n * 144 + k], pa); split3(h[k * 16 + n], pb); }
      for (int i = 0; i < 3; ++i) { A[i][s][j] = __builtin_bit_cast(__bf16, pa[i]); B[i][s][j] = __builtin_bit_cast(__bf16, pb[i]); }
    }
  const int order[9][2] = {{2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}, {2, 1}, {1, 2}, {2, 2}};   // first 6 = the "x6" set, small terms first
  for (int variant = 1; variant < 4; ++variant) {
    f32x4 c = {0, 0, 0, 0};
    const int lo = variant == 1 ? 3 : 0, hi = variant == 3 ? 9 : 6;   // x3: (1,0),(0,1),(0,0); x6; x9
    if (variant == 3)
      for (int q = 6; q < 9; ++q)
        for (int s = 0; s < 5; ++s) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[order[q][0]][s], B[order[q][1]][s], c, 0, 0, 0);
    for (int q = lo; q < (hi > 6 ? 6 : hi); ++q)
      for (int s = 0; s < 5; ++s) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[order[q][0]][s], B[order[q][1]][s], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(((size_t)t * 4 + variant) * 16 + 4 * g + r) * 16 + n] = c[r];
  }
  {   // x6 with the last 16 k on v_mfma_f32_16x16x16_bf16 in its OWN accumulator (overwrites the x3 slot)
    f32x4 c = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
    s16x4 At[3], Bt[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) { At[i][j] = 0; Bt[i][j] = 0; }
    for (int j = 0; j < 4; ++j) {
      unsigned short pa[3], pb[3];
      split3(w[n * 144 + 128 + 4 * g + j], pa); split3(h[(128 + 4 * g + j) * 16 + n], pb);
      for (int i = 0; i < 3; ++i) { At[i][j] = (short)pa[i]; Bt[i][j] = (short)pb[i]; }
    }
    for (int q = 0; q < 6; ++q) {
      for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[order[q][0]][s], B[order[q][1]][s], c, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(At[order[q][0]], Bt[order[q][1]], c2, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) out[(((size_t)t * 4 + 1) * 16 + 4 * g + r) * 16 + n] = c[r] + c2[r];
  }
  {   // everything on v_mfma_f32_16x16x16_bf16, 10 half steps (overwrites the x9 slot)
    f32x4 c = {0, 0, 0, 0};
    for (int q = 0; q < 6; ++q)
      for (int s = 0; s < 5; ++s)
        for (int hf = 0; hf < 2; ++hf) {
          s16x4 a4, b4;
          for (int j = 0; j < 4; ++j) { a4[j] = __builtin_bit_cast(short, A[order[q][0]][s][4 * hf + j]); b4[j] = __builtin_bit_cast(short, B[order[q][1]][s][4 * hf + j]); }
          c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c, 0, 0, 0);
        }
    for (int r = 0; r < 4; ++r) out[(((size_t)t * 4 + 3) * 16 + 4 * g + r) * 16 + n] = c[r];
  }
}

// ------------------------------------------------------------------ part B
// Wp: [n_tiles][3 pieces][5 steps][64 lanes][8 bf16] = 15 KiB per tile (lane-linear fragments)
#define TILE_BYTES (15 * 1024)
template <int NW, int NBK, int MODE, int OVH>   // MODE 0: A from LDS, no loads at all; 1: LDS-DMA double buffer (one stream per CU); 2: per-wave buffer loads (L1)
__global__ __launch_bounds__(64 * NW, 1) void k_sweep(const char* Wp, float* out, unsigned long long* cyc, int tiles, int n_tiles_w) {
  __shared__ __attribute__((aligned(16))) char ring[2][TILE_BYTES];
  __shared__ __attribute__((aligned(16))) float xs[NW * 16 * NBK * 24];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 B[NBK][3][5];
  for (int b = 0; b < NBK; ++b)
    for (int i = 0; i < 3; ++i)
      for (int s = 0; s < 5; ++s) {
        u32x4 v;
        for (int q = 0; q < 4; ++q) { unsigned h = (lane * 2654435761u) ^ ((((b * 3 + i) * 5 + s) * 4 + q + 1) * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; v[q] = (h & 0x3f803f80u) | 0x30003000u; }
        B[b][i][s] = __builtin_bit_cast(bf16x8, v);
      }
  for (int i = threadIdx.x; i < NW * 16 * NBK * 24; i += blockDim.x) xs[i] = 0.01f * (i % 97);
  for (int i = threadIdx.x * 16; i < TILE_BYTES; i += blockDim.x * 16) *reinterpret_cast<u32x4*>(&ring[0][i]) = *reinterpret_cast<const u32x4*>(Wp + i);
  const float* xl = xs + (wave * 16 * NBK + (lane & 15)) * 24;
  float oacc[NBK][3] = {};
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, n_tiles_w * TILE_BYTES, 0x00020000);
  bf16x8 Ar[3][5];
  if (MODE == 2)
    for (int i = 0; i < 3; ++i)
      for (int s = 0; s < 5; ++s) Ar[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, (i * 5 + s) * 1024, 0));
  __syncthreads();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < tiles; ++t) {
    const int tn = (t + 1) % n_tiles_w;
    const char* cur = ring[MODE == 1 ? (t & 1) : 0];
    if (MODE == 1) {
      // the tile computed two iterations ago is dead: every wave refills a share of its buffer with the NEXT tile
      for (int f = wave; f < 15; f += NW)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(Wp + (size_t)tn * TILE_BYTES + f * 1024 + lane * 16),
                                         (void __attribute__((address_space(3)))*)(&ring[(t + 1) & 1][f * 1024]), 16, 0, 0);
    }
    f32x4 acc[NBK];
    for (int b = 0; b < NBK; ++b) acc[b] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      bf16x8 a0, a1, a2;
      if (MODE == 2) { a0 = Ar[0][s]; a1 = Ar[1][s]; a2 = Ar[2][s]; }
      else {
        a0 = *reinterpret_cast<const bf16x8*>(cur + (0 * 5 + s) * 1024 + lane * 16);
        a1 = *reinterpret_cast<const bf16x8*>(cur + (1 * 5 + s) * 1024 + lane * 16);
        a2 = *reinterpret_cast<const bf16x8*>(cur + (2 * 5 + s) * 1024 + lane * 16);
      }
#pragma unroll
      for (int b = 0; b < NBK; ++b) {
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, B[b][0][s], acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, B[b][1][s], acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, B[b][2][s], acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, B[b][0][s], acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, B[b][1][s], acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, B[b][0][s], acc[b], 0, 0, 0);
      }
      if (MODE == 2) {
        __builtin_amdgcn_sched_barrier(0);
        for (int i = 0; i < 3; ++i) Ar[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, tn * TILE_BYTES + (i * 5 + s) * 1024, 0));
      }
    }
    if (OVH) {   // k_conv-like tail of a tile: per edge block 3 LDS reads and ~16 FMAs
#pragma unroll
      for (int b = 0; b < NBK; ++b) {
        const f32x4 v = acc[b];
        const f32x4* x4 = reinterpret_cast<const f32x4*>(xl + 16 * b * 24 + 4 * (t & 1));
        const f32x4 xa = x4[0], xb = x4[1], xc = x4[2];
        const float z0 = v[0] * xa[0] + v[1] * xa[3] + v[2] * xb[2] + v[3] * xc[1];
        const float z1 = v[0] * xa[1] + v[1] * xb[0] + v[2] * xb[3] + v[3] * xc[2];
        const float z2 = v[0] * xa[2] + v[1] * xb[1] + v[2] * xc[0] + v[3] * xc[3];
        oacc[b][0] += z1 * xa[2] - z2 * xa[1]; oacc[b][1] += z2 * xa[0] - z0 * xa[2]; oacc[b][2] += z0 * xa[1] - z1 * xa[0];
      }
    } else {
      for (int b = 0; b < NBK; ++b) oacc[b][0] += acc[b][0] + acc[b][1] + acc[b][2] + acc[b][3];
    }
    if (MODE == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float tot = 0.f;
  for (int b = 0; b < NBK; ++b) tot += oacc[b][0] + oacc[b][1] + oacc[b][2];
  out[blockIdx.x * 64 * NW + threadIdx.x] = tot;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = r1 - r0;
}

// ------------------------------------------------------------------ part C
template <int KIND>
__global__ void k_issue(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x;
  u32x4 v = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u};
  bf16x8 a = __builtin_bit_cast(bf16x8, v), b = a;
  s16x4 a4 = {(short)0x3f80, (short)0x3f80, (short)0x3f00, (short)(0x3f80 + lane)}, b4 = a4;
  f32x4 c[4] = {};
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (KIND == 0) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[q], 0, 0, 0);
      else c[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c[q], 0, 0, 0);
    }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  out[lane] = c[0][0] + c[1][0] + c[2][0] + c[3][0];
  if (lane == 0) { cyc[0] = c1 - c0; cyc[1] = r1 - r0; }
}

int main() {
  // ---- part A
  {
    const int T = 512;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> W((size_t)T * 16 * 144), H((size_t)T * 144 * 16);
    for (auto& x : W) x = 0.12f * nd(rng);
    for (auto& x : H) { float v = nd(rng); x = v > 0 ? v : 0.f; }   // relu outputs
    float *dW, *dH, *dO;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dH, H.size() * 4); hipMalloc(&dO, (size_t)T * 4 * 256 * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_acc, dim3(T), dim3(64), 0, 0, dW, dH, dO);
    std::vector<float> O((size_t)T * 4 * 256);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
    const char* names[5] = {"native v_mfma_f32_16x16x4_f32", "x6, x16 tail in own accumulator", "bf16 split, 6 products", "x6, all on 16x16x16", "host float fma chain"};
    double mx[5] = {}, sq[5] = {};
    for (int t = 0; t < T; ++t)
      for (int r = 0; r < 16; ++r)
        for (int n = 0; n < 16; ++n) {
          double ref = 0, mag = 0;
          float f = 0.f;
          for (int k = 0; k < 144; ++k) {
            const double p = (double)W[((size_t)t * 16 + r) * 144 + k] * (double)H[((size_t)t * 144 + k) * 16 + n];
            ref += p; mag += fabs(p);
            f = fmaf(W[((size_t)t * 16 + r) * 144 + k], H[((size_t)t * 144 + k) * 16 + n], f);
          }
          for (int v = 0; v < 5; ++v) {
            const double got = v < 4 ? O[(((size_t)t * 4 + v) * 16 + r) * 16 + n] : f;
            const double e = fabs(got - ref) / mag;
            mx[v] = fmax(mx[v], e); sq[v] += e * e;
          }
        }
    printf("part A: |err| / sum_k |W h|  over %d outputs (fp64 reference); fp32 eps = %.3e\n", T * 256, ldexp(1.0, -24));
    for (int v = 0; v < 5; ++v) printf("  %-34s max %.3e   rms %.3e\n", names[v], mx[v], sqrt(sq[v] / (T * 256.0)));
  }
  // ---- part C
  {
    float* out; unsigned long long* cyc; hipMalloc(&out, 256); hipMalloc(&cyc, 16);
    for (int kind = 0; kind < 2; ++kind) {
      const int iters = 20000;
      if (kind == 0) { hipLaunchKernelGGL(k_issue<0>, dim3(1), dim3(64), 0, 0, out, cyc, iters); }
      else { hipLaunchKernelGGL(k_issue<1>, dim3(1), dim3(64), 0, 0, out, cyc, iters); }
      hipDeviceSynchronize();
      unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
      printf("part C: %s  %.2f shader cycles, %.2f ns per MFMA (one wave, 4 accumulators)\n", kind == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x16_bf16",
             (double)h[0] / (4.0 * iters), (double)h[1] * 10.0 / (4.0 * iters));
    }
  }
  // ---- part B
  if (!getenv("SKIP_B")) {
    const int n_tiles_w = 486, tiles = 486 * 2;
    std::vector<unsigned> Wp((size_t)n_tiles_w * TILE_BYTES / 4);
    std::mt19937 rng(11);
    for (auto& x : Wp) x = (rng() & 0xbfffbfffu) | 0x30003000u;   // random bf16 pairs of moderate magnitude
    char* dW; hipMalloc(&dW, Wp.size() * 4); hipMemcpy(dW, Wp.data(), Wp.size() * 4, hipMemcpyHostToDevice);
    float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
#define RUN(NW, NBK, MODE, OVH, what) do { \
      hipLaunchKernelGGL((k_sweep<NW, NBK, MODE, OVH>), dim3(256), dim3(64 * NW), 0, 0, dW, out, cyc, 40, n_tiles_w); hipDeviceSynchronize(); \
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0); \
      hipLaunchKernelGGL((k_sweep<NW, NBK, MODE, OVH>), dim3(256), dim3(64 * NW), 0, 0, dW, out, cyc, tiles, n_tiles_w); \
      hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1); \
      hipError_t err = hipGetLastError(); \
      const double flop = 256.0 * NW * NBK * 16.0 * 16.0 * 144.0 * 2.0 * tiles; \
      printf("  %-72s %8.3f ms  %7.1f fp32-equivalent TFLOP/s  (%6.1f ns / tile / CU)%s\n", what, ms, flop / (ms * 1e-3) * 1e-12, ms * 1e6 / tiles, err ? "  LAUNCH ERROR" : ""); } while (0)
    printf("part B: 256 workgroups, %d tiles each; fp32 MFMA path today: 138.8 TFLOP/s\n", tiles);
    RUN(8, 2, 0, 0, "8 waves x 32 edges, A from LDS (no loads), no tail");
    RUN(8, 2, 0, 1, "8 waves x 32 edges, A from LDS (no loads), contraction tail");
    RUN(8, 2, 1, 0, "8 waves x 32 edges, LDS-DMA double buffer, no tail");
    RUN(8, 2, 1, 1, "8 waves x 32 edges, LDS-DMA double buffer, contraction tail");
    RUN(8, 2, 2, 1, "8 waves x 32 edges, per-wave buffer loads (L1), contraction tail");
    RUN(4, 4, 0, 1, "4 waves x 64 edges, A from LDS (no loads), contraction tail");
    RUN(4, 4, 1, 1, "4 waves x 64 edges, LDS-DMA double buffer, contraction tail");
    RUN(4, 4, 2, 1, "4 waves x 64 edges, per-wave buffer loads (L1), contraction tail");
    RUN(4, 3, 1, 1, "4 waves x 48 edges, LDS-DMA double buffer, contraction tail");
    RUN(8, 3, 1, 1, "8 waves x 48 edges, LDS-DMA double buffer, contraction tail");
  }
  return 0;
}
