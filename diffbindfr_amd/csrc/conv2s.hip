// Fused tensor-product convolution, third generation (gfx950): k_conv2's persistent edge-owner structure with the radial
// MLP's big GEMM (hidden layer -> per-edge tensor-product weights, 144 x W, 97-99 % of the conv's arithmetic) moved from the
// fp32 matrix instruction (v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD) to the bf16 one (v_mfma_f32_16x16x32_bf16, 16x the rate)
// WITHOUT giving up fp32 results:
//
//   * every fp32 operand is cut into three bf16 pieces, a = a1 + a2 + a3 (round-to-nearest each time; 3 x 8 significand bits
//     hold all 24 bits of an fp32 number, so the sum is exact).  W2 is cut once at model creation (api.cpp pack_conv2), the
//     hidden layer h is cut in registers right after its ReLU;
//   * a product a b = sum_ij a_i b_j; each a_i b_j (8 x 8 bits) is exact in the fp32 accumulator of the matrix pipe.  The six
//     partial products with i + j <= 4 are evaluated, smallest first; the three dropped ones are below 2^-23 |a b|.  Measured
//     against an fp64 reference (tools/exp/split_bf16.hip, K = 144, ReLU-like h): max error 1.05e-7 of sum|w h| (rms 9.7e-9),
//     vs 3.13e-7 (rms 2.8e-8) for the native fp32 MFMA chain this replaces and for a host fmaf chain -- the split form is
//     the MORE accurate of the two, because it rounds 30 times per dot product instead of 144;
//   * the reduction index is padded 144 -> 160 (5 k-steps of 32) and permuted so that the hidden layer's MFMA result
//     registers are, piece by piece, the B operand of the W2 tiles: step s, lane group g, slot j holds hidden unit
//     16 (2 s + (j >> 2)) + 4 g + (j & 3) -- no transpose, no LDS, as in k_conv2;
//   * the first GEMM (144 x 144, 2-5 % of the arithmetic) stays on the native fp32 instruction.
//
// 30 bf16 MFMAs (16 cycles each) replace 36 fp32 ones (32 cycles each) per 16 x 16 weight tile: 2.4x less matrix-pipe time.
// The W2 fragment stream grows from 9 to 15 KiB per tile (three pieces, padded k), which is why the edge-owner structure
// (one stream per CU through L1, shared by the 8 waves of the workgroup) is the one that carries it.
// Everything after the accumulators -- bias as initial value, closed-form Clebsch-Gordan contraction, channel-owner
// accumulation, unit queue, tail split -- is k_conv2's (conv2.hip), the C/D register layout of the two instructions being
// the same.  Results are bitwise independent of the unit -> workgroup assignment and of the split.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define C3_TILE_BYTES 13824         // [3 pieces][4 k-steps of 32][64 lanes][8 bf16] + [3 pieces][64 lanes][4 bf16] (the last 16 k)
#define C3_TAIL_OFF 12288

// two fp32 values -> three words of packed bf16 pairs (low half = x0's piece): x = p0 + p1 + p2 exactly, every piece rounded
// to nearest even by v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float x0, float x1) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, bf16x2));
}
__device__ __forceinline__ void split3x2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pk_bf16(x0, x1);
  float r0 = x0 - __builtin_bit_cast(float, p0 << 16), r1 = x1 - __builtin_bit_cast(float, p0 & 0xffff0000u);
  p1 = pk_bf16(r0, r1);
  r0 -= __builtin_bit_cast(float, p1 << 16); r1 -= __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = pk_bf16(r0, r1);
}

#define C2_XLD 124                                   // LDS x row: 120 floats + 4 (odd multiple of 4: 16 rows -> 16 distinct 16-B slots)
#define C2_WAVE_FLOATS (32 * C2_XLD + 32 * 10 + 32 * 8 + 32)   // x rows | harmonics | l=2 matrix | gather indices

// RB 2: s_barrier at every tile start (keeps the 8 waves on the same W2 tile: the L1 window is two 15-KiB tiles);
// ABL (developer, wrong results): 1 no contraction, 2 no W2 fragment re-load, 4 no hidden layer
template <int NW, int RB = 0, int ABL = 0>
__global__ __launch_bounds__(64 * NW, 2) void k_conv2s(Conv2Args a) {
  constexpr int K = 144, KT = 9;
  constexpr int EPB = 32 * NW;                       // edges per block (unit)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_unit[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  float* xs = lds + wave * C2_WAVE_FLOATS;           // [32][C2_XLD]
  float* shs = xs + 32 * C2_XLD;                     // [32][10]
  float* ms = shs + 32 * 10;                         // [32][8]
  int* s_gth = reinterpret_cast<int*>(ms + 32 * 8);  // [32]

  // ---- the launch's unit list (every workgroup derives it from the device-side edge counts)
  int nb0 = 0, nb1 = 0, nb2 = 0, nb3 = 0;
  {
    auto blocks = [&](int c) { return c < a.n_conv ? (min(*a.c[c].n_edges, a.c[c].max_edges) + EPB - 1) / EPB : 0; };
    nb0 = blocks(0); nb1 = blocks(1); nb2 = blocks(2); nb3 = blocks(3);
  }
  const int N = nb0 + nb1 + nb2 + nb3;
  const int n_wg = gridDim.x;
  const int full = (N / n_wg) * n_wg, rem = N - full;
  int si = 0;
  if (rem > 0 && !a.no_split) { const int q = n_wg / rem; si = q >= 8 ? 3 : q >= 4 ? 2 : q >= 2 ? 1 : 0; }
  const int total = full + (rem << si);

  for (int it = 0;; ++it) {
    if (tid == 0) s_unit[it & 1] = atomicAdd(a.queue, 1);
    __syncthreads();
    const int u = s_unit[it & 1];
    if (u >= total) break;
    int blk, part = 0, psi = 0;
    if (u < full) blk = u;
    else { const int v = u - full; blk = full + (v >> si); part = v & ((1 << si) - 1); psi = si; }
    int c = 0;
    if (blk >= nb0) { blk -= nb0; c = 1; if (blk >= nb1) { blk -= nb1; c = 2; if (blk >= nb2) { blk -= nb2; c = 3; } } }
    const Conv2Desc& d = a.c[c];
    const int E = min(*d.n_edges, d.max_edges);
    const int e0 = blk * EPB + 32 * wave;
    if (RB == 0 && e0 >= E) continue;                           // this wave has no edge in the block (the loop-top barrier is still reached)
    const int D_in = d.w.D_in, D_out = d.w.D_out;

    // ---- my two edges (block b, column n), clamped; gather indices
    int ev[2], gthv[2];
    bool valid[2];
    const float* r0[2]; const float* r1[2]; const float* r2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int e = e0 + 16 * b + n;
      valid[b] = e < E;
      ev[b] = min(e, E - 1);
      gthv[b] = d.gth[ev[b]];
      r0[b] = d.emb + (size_t)ev[b] * NS;
      r1[b] = d.tab1 + (size_t)d.idx1[ev[b]] * d.ld1;
      r2[b] = d.tab2 + (size_t)d.idx2[ev[b]] * d.ld2;
    }
    if (g == 0) { s_gth[n] = gthv[0]; s_gth[16 + n] = gthv[1]; }
    if (lane < 32) {   // harmonics of the wave's 32 edges + the closed form of the 1 x 2 -> 1 coupling (so3_host.cpp)
      const int e = min(e0 + lane, E - 1);
      const float* sp = d.sh + (size_t)e * SH_LD;
      float s[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) { s[k] = sp[k]; shs[lane * 10 + k] = s[k]; }
      const float r3 = 1.7320508075688772f;
      float* m = ms + lane * 8;
      m[0] = -s[6] - r3 * s[8]; m[1] = r3 * s[5]; m[2] = r3 * s[4]; m[3] = 2.f * s[6]; m[4] = r3 * s[7]; m[5] = -s[6] + r3 * s[8];
    }
    // ---- hidden layer on the matrix cores (native fp32 instruction), all in registers, one edge block at a time:
    // h = relu(W1 a + b1); the radial-MLP input row [emb | tab1 | tab2] of the lane's edge goes straight into B-operand
    // registers (k-step q of lane group g of 16-group s4 takes k = 16 s4 + 4 g + q, api.cpp pack_conv);
    // D[row = hidden unit, col = edge] -> lane (g, n) ends up with h[16 m + 4 g + r][edge n], r = 0..3, which it cuts
    // into bf16 pieces and files as slots 4 (m & 1) + r of k-step m >> 1 of the W2 tiles' B operand
    const int vW = lane * 16;
    u32x4 Bh[2][3][4];                                 // h pieces: [edge block][piece][k-step of 32] = 8 bf16 each, 96 VGPRs
    u32x2 Bt[2][3];                                    // ... and of the last 16 k (v_mfma_f32_16x16x16_bf16): 4 bf16 each, 12 VGPRs
    {
      const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.W1p, 0, KT * KT * 1024, 0x00020000);
#pragma unroll
      for (int b = 0; b < ((ABL & 4) ? 0 : 2); ++b) {
        f32x4 Ba[KT];
#pragma unroll
        for (int s4 = 0; s4 < KT; ++s4) {
          const float* src = s4 < 3 ? r0[b] : s4 < 6 ? r1[b] : r2[b];
          Ba[s4] = *reinterpret_cast<const f32x4*>(src + 16 * (s4 % 3) + 4 * g);
        }
        // W1 fragments through a ring of 4 registers sets, requested 4 k-groups (16 MFMAs) ahead: a resident copy of a whole
        // row tile (36 VGPRs) next to the growing piece file makes the allocator spill into the W2 sweep
        constexpr int RING = 4;
        f32x4 A1[RING];
#pragma unroll
        for (int it = 0; it < RING; ++it) A1[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW1, vW, it * 1024, 0));
#pragma unroll
        for (int m = 0; m < KT; ++m) {
          f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};   // two chains over alternate k-steps (dependent-issue latency)
#pragma unroll
          for (int s4 = 0; s4 < KT; ++s4) {
            const int it = m * KT + s4;
            const f32x4 av = A1[it % RING];
            h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], Ba[s4][0], h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], Ba[s4][1], h1, 0, 0, 0);
            h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], Ba[s4][2], h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], Ba[s4][3], h1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (it + RING < KT * KT) A1[it % RING] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW1, vW, (it + RING) * 1024, 0));
          }
          const f32x4 b1v = *reinterpret_cast<const f32x4*>(d.w.b1 + 16 * m + 4 * g);
          float hv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = fmaxf((h0[r] + h1[r]) + b1v[r], 0.f);
          unsigned pc[3][2];
          split3x2(hv[0], hv[1], pc[0][0], pc[1][0], pc[2][0]);
          split3x2(hv[2], hv[3], pc[0][1], pc[1][1], pc[2][1]);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            if (m < 8) { Bh[b][i][m >> 1][2 * (m & 1)] = pc[i][0]; Bh[b][i][m >> 1][2 * (m & 1) + 1] = pc[i][1]; }
            else { Bt[b][i][0] = pc[i][0]; Bt[b][i][1] = pc[i][1]; }
          }
        }
      }
    }
    // ---- x[gth] rows of the wave's 32 edges into its LDS region (columns [0, min(D_in, 120)))
    __builtin_amdgcn_wave_barrier();
    {
      const int d4 = min(D_in, 120) >> 2;
      for (int i = lane; i < 32 * d4; i += 64) {
        const int el = i / d4, c4 = i - el * d4;
        *reinterpret_cast<f32x4*>(xs + el * C2_XLD + 4 * c4) = *reinterpret_cast<const f32x4*>(d.x + (size_t)s_gth[el] * d.ldx + 4 * c4);
      }
    }
    __builtin_amdgcn_wave_barrier();
    int x_phase = 0;

    // ---- the W2 row tiles of this part, run by run (channel-owner order, api.cpp pack_conv2)
    const int r_begin = d.w.part_run[psi][part], r_end = d.w.part_run[psi][part + 1];
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.W2s, 0, d.w.n_tiles * C3_TILE_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.b2q, 0, d.w.n_tiles * 64, 0x00020000);
    const int vB = g * 16;
    bf16x8 A[3][4];                                    // W2 pieces of the current tile: 48 VGPRs ...
    s16x4 At[3];                                       // ... + 6 for the last 16 k
    const int vT = lane * 8;
    f32x4 bias_n = {0.f, 0.f, 0.f, 0.f};
    int t_last = 0;
    if (r_begin < r_end) {
      const RunDesc rl = d.w.runs[r_end - 1];
      t_last = (rl.tile0_n & 0xfffff) + (rl.tile0_n >> 20) - 1;
      const int t0 = d.w.runs[r_begin].tile0_n & 0xfffff;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          A[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, vW, t0 * C3_TILE_BYTES + (i * 4 + s) * 1024, 0));
#pragma unroll
      for (int i = 0; i < 3; ++i)
        At[i] = __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(rW, vT, t0 * C3_TILE_BYTES + C3_TAIL_OFF + i * 512, 0));
      bias_n = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, vB, t0 * 64, 0));
    }
    float oacc[2][3];
#pragma unroll
    for (int b = 0; b < 2; ++b) oacc[b][0] = oacc[b][1] = oacc[b][2] = 0.f;
    const float* xs_lane = xs + n * C2_XLD;          // + 16 b C2_XLD per edge block
    const float* sh_lane = shs + n * 10;
    // the two waves of a SIMD run the same code from the same start: left alone they stay IN phase (both in their MFMA
    // burst, then both in their contraction, the matrix pipe idle).  The second-dispatched half starts its tiles a
    // fraction of a tile late; the offset then persists (a wave that contracts lets its partner run at full rate)
    if (wave >= NW / 2) for (int i = 0; i < a.skew; ++i) __builtin_amdgcn_s_sleep(8);
    for (int r = r_begin; r < r_end; ++r) {
      const RunDesc rd = d.w.runs[r];
      const int tile0 = rd.tile0_n & 0xfffff, nt = rd.tile0_n >> 20;
      const int type = rd.meta & 15, flags = (rd.meta >> 4) & 3, sh_off = (rd.meta >> 8) & 15, x_step = (rd.meta >> 12) & 0xff;
      const int xph = (rd.meta >> 20) & 1;
      const int xo = (rd.x_off4 >> (8 * g)) & 0xff, oo = (rd.o_off4 >> (8 * g)) & 0xff;
      if (xph != x_phase) {   // second output half: the scalar slot [0,48) now holds the 48x0o inputs x[120:168]
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < 32 * 12; i += 64) {
          const int el = i / 12, c4 = i - el * 12;
          *reinterpret_cast<f32x4*>(xs + el * C2_XLD + 4 * c4) = *reinterpret_cast<const f32x4*>(d.x + (size_t)s_gth[el] * d.ldx + 120 + 4 * c4);
        }
        __builtin_amdgcn_wave_barrier();
        x_phase = xph;
      }
      if (flags & 1) {
#pragma unroll
        for (int b = 0; b < 2; ++b) oacc[b][0] = oacc[b][1] = oacc[b][2] = 0.f;
      }
      auto run = [&](auto type_c) {
        constexpr int TYPE = decltype(type_c)::value;
        constexpr bool VIN = !(TYPE == PT_SS || TYPE == PT_SV);
        constexpr bool VOUT = !(TYPE == PT_SS || TYPE == PT_VVS);
        constexpr int NSV = (TYPE == PT_SS || TYPE == PT_VS) ? 1 : (TYPE == PT_VTV ? 0 : 3);
        float S[2][NSV ? NSV : 1];
        if (TYPE != PT_VTV) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const float* sp = sh_lane + 160 * b + sh_off;
#pragma unroll
            for (int k = 0; k < NSV; ++k) S[b][k] = sp[k];
          }
        }
        const float* xp = xs_lane + xo;
        for (int i = 0; i < nt; ++i, xp += x_step) {
          const int t = tile0 + i;
          const int tn = t < t_last ? t + 1 : t;
          f32x4 acc[2];
          if (RB >= 2) __builtin_amdgcn_s_barrier();
          acc[0] = bias_n; acc[1] = bias_n;
          bias_n = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, vB, tn * 64, 0));
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bf16x8 a0 = A[0][s], a1 = A[1][s], a2 = A[2][s];
            // the six partial products with i + j <= 4, smallest first; two independent accumulator chains
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, __builtin_bit_cast(bf16x8, Bh[0][0][s]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, __builtin_bit_cast(bf16x8, Bh[1][0][s]), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, Bh[0][1][s]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, Bh[1][1][s]), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, Bh[0][2][s]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, Bh[1][2][s]), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, Bh[0][0][s]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, Bh[1][0][s]), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, Bh[0][1][s]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, Bh[1][1][s]), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, Bh[0][0][s]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, Bh[1][0][s]), acc[1], 0, 0, 0);
            // pin the re-loads BEHIND the last MFMA that reads these registers (one fragment set, no spills)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
              if (!(ABL & 2)) A[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, vW, tn * C3_TILE_BYTES + (i * 4 + s) * 1024, 0));
          }
          if (ABL & 8) {   // developer: the last 16 k on the x32 instruction with zero-padded operands
            bf16x8 ta[3], tb[2][3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const u32x2 av = __builtin_bit_cast(u32x2, At[i]);
              ta[i] = __builtin_bit_cast(bf16x8, (u32x4){av[0], av[1], 0u, 0u});
#pragma unroll
              for (int b = 0; b < 2; ++b) tb[b][i] = __builtin_bit_cast(bf16x8, (u32x4){Bt[b][i][0], Bt[b][i][1], 0u, 0u});
            }
            const int oi[6] = {2, 1, 0, 1, 0, 0}, oj[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
              for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta[oi[q]], tb[b][oj[q]], acc[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
              At[i] = __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(rW, vT, tn * C3_TILE_BYTES + C3_TAIL_OFF + i * 512, 0));
          } else if (ABL & 16) {   // developer: the x16 chain in its own accumulators
            const s16x4 a0 = At[0], a1 = At[1], a2 = At[2];
            f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#define TB(b, i) __builtin_bit_cast(s16x4, Bt[b][i])
            t0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2, TB(0, 0), t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2, TB(1, 0), t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, TB(0, 1), t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, TB(1, 1), t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(0, 2), t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(1, 2), t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, TB(0, 0), t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, TB(1, 0), t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(0, 1), t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(1, 1), t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(0, 0), t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(1, 0), t1, 0, 0, 0);
#undef TB
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
              At[i] = __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(rW, vT, tn * C3_TILE_BYTES + C3_TAIL_OFF + i * 512, 0));
            acc[0] += t0; acc[1] += t1;
          } else {   // k = 128..143 on v_mfma_f32_16x16x16_bf16 (two-register operands).
            // Hazard (found the hard way, MI355X + ROCm 7.2): an x16 MFMA that takes as SrcC an accumulator an x32 MFMA has just
            // written reads stale data -- hipcc pads nothing between the two opcodes (wrong, timing-dependent results; the same
            // chain is clean when either opcode is used alone, tools/exp/mfma_x16_chain.hip).  16 wait states and the acc[0] chain
            // first (its last x32 write is six MFMAs old) put any pass count behind us; costs < 1 % of a tile.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15");
            __builtin_amdgcn_sched_barrier(0);
            const s16x4 a0 = At[0], a1 = At[1], a2 = At[2];
#define TB(b, i) __builtin_bit_cast(s16x4, Bt[b][i])
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2, TB(b, 0), acc[b], 0, 0, 0);
              acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, TB(b, 1), acc[b], 0, 0, 0);
              acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(b, 2), acc[b], 0, 0, 0);
              acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, TB(b, 0), acc[b], 0, 0, 0);
              acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(b, 1), acc[b], 0, 0, 0);
              acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(b, 0), acc[b], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
#undef TB
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
              if (!(ABL & 2)) At[i] = __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(rW, vT, tn * C3_TILE_BYTES + C3_TAIL_OFF + i * 512, 0));
          }
          if (ABL & 1) {
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
            continue;
          }
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const f32x4 v = acc[b];
            const f32x4* x4 = reinterpret_cast<const f32x4*>(xp + 16 * b * C2_XLD);
            const f32x4 xa = x4[0];
            if (!VIN) {
              const float z = v[0] * xa[0] + v[1] * xa[1] + v[2] * xa[2] + v[3] * xa[3];
              oacc[b][0] += z * S[b][0];
              if (TYPE == PT_SV) { oacc[b][1] += z * S[b][1]; oacc[b][2] += z * S[b][2]; }
            } else {
              const f32x4 xb = x4[1], xc = x4[2];   // [u0..u0+3][3] = 12 consecutive floats
              const float z0 = v[0] * xa[0] + v[1] * xa[3] + v[2] * xb[2] + v[3] * xc[1];
              const float z1 = v[0] * xa[1] + v[1] * xb[0] + v[2] * xb[3] + v[3] * xc[2];
              const float z2 = v[0] * xa[2] + v[1] * xb[1] + v[2] * xc[0] + v[3] * xc[3];
              if (TYPE == PT_VS) {
                oacc[b][0] += z0 * S[b][0]; oacc[b][1] += z1 * S[b][0]; oacc[b][2] += z2 * S[b][0];
              } else if (TYPE == PT_VVS) {
                oacc[b][0] += z0 * S[b][0] + z1 * S[b][1] + z2 * S[b][2];
              } else if (TYPE == PT_VVV) {
                oacc[b][0] += z1 * S[b][2] - z2 * S[b][1];
                oacc[b][1] += z2 * S[b][0] - z0 * S[b][2];
                oacc[b][2] += z0 * S[b][1] - z1 * S[b][0];
              } else {   // PT_VTV: symmetric traceless matrix of the l=2 harmonics
                const f32x4 ma = *reinterpret_cast<const f32x4*>(ms + (16 * b + n) * 8);
                const float2 mb = *reinterpret_cast<const float2*>(ms + (16 * b + n) * 8 + 4);
                oacc[b][0] += ma[0] * z0 + ma[1] * z1 + ma[2] * z2;
                oacc[b][1] += ma[1] * z0 + ma[3] * z1 + mb.x * z2;
                oacc[b][2] += ma[2] * z0 + mb.x * z1 + mb.y * z2;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if (flags & 2) {   // last run of the channel group: this lane owns msg[e][oo .. oo + (VOUT ? 3 : 1))
          if (oo < D_out) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
              if (valid[b]) {
                float* op = d.msg + (size_t)ev[b] * D_out + oo;
                op[0] = oacc[b][0];
                if (VOUT) { op[1] = oacc[b][1]; op[2] = oacc[b][2]; }
              }
          }
        }
      };
      switch (type) {
        case PT_SS: run(std::integral_constant<int, PT_SS>{}); break;
        case PT_SV: run(std::integral_constant<int, PT_SV>{}); break;
        case PT_VS: run(std::integral_constant<int, PT_VS>{}); break;
        case PT_VVS: run(std::integral_constant<int, PT_VVS>{}); break;
        case PT_VVV: run(std::integral_constant<int, PT_VVV>{}); break;
        default: run(std::integral_constant<int, PT_VTV>{}); break;
      }
    }
  }
  // ---- the last workgroup to leave re-arms the queue for the next launch
  if (tid == 0) {
    const int dn = atomicAdd(a.queue + 1, 1);
    if (dn == n_wg - 1) { a.queue[0] = 0; a.queue[1] = 0; __threadfence(); }
  }
}

void launch_conv2s(const Conv2Args& a, hipStream_t st) {
  static int n_cu = 0;
  static int no_split = getenv("DBFR_CONV2_NOSPLIT") ? atoi(getenv("DBFR_CONV2_NOSPLIT")) : 0;
  static int skew = getenv("DBFR_CONV2_SKEW") ? atoi(getenv("DBFR_CONV2_SKEW")) : 0;
  constexpr int NW = 8;
  const size_t lds = (size_t)NW * C2_WAVE_FLOATS * sizeof(float);
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 0, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 0, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  static int var = getenv("DBFR_CONV2S_VAR") ? atoi(getenv("DBFR_CONV2S_VAR")) : 0;   // developer: 10 RB + ABL
  Conv2Args b = a;
  b.skew = skew;
  b.run_barrier = 0;
  b.no_split = no_split;
  const dim3 gr(n_cu), bl(64 * NW);
  switch (var) {
    case 20: hipLaunchKernelGGL((k_conv2s<NW, 2, 0>), gr, bl, lds, st, b); break;
    case 22: hipLaunchKernelGGL((k_conv2s<NW, 2, 2>), gr, bl, lds, st, b); break;
    case 1: hipLaunchKernelGGL((k_conv2s<NW, 0, 1>), gr, bl, lds, st, b); break;
    case 2: hipLaunchKernelGGL((k_conv2s<NW, 0, 2>), gr, bl, lds, st, b); break;
    case 3: hipLaunchKernelGGL((k_conv2s<NW, 0, 3>), gr, bl, lds, st, b); break;
    case 4: hipLaunchKernelGGL((k_conv2s<NW, 0, 4>), gr, bl, lds, st, b); break;
    case 8: hipLaunchKernelGGL((k_conv2s<NW, 0, 8>), gr, bl, lds, st, b); break;
    case 16: hipLaunchKernelGGL((k_conv2s<NW, 0, 16>), gr, bl, lds, st, b); break;
    case 32: hipLaunchKernelGGL((k_conv2s<NW, 0, 32>), gr, bl, lds, st, b); break;
    default: hipLaunchKernelGGL((k_conv2s<NW, 0, 0>), gr, bl, lds, st, b); break;
  }
}
