// Fused tensor-product convolution, second generation (gfx950): ONE W2 stream per compute unit.
//
// k_conv (conv.hip) gives every wave of a 4-wave workgroup a quarter of the W2 row tiles for 48 edges: a CU then runs
// eight distinct weight streams (4 waves x 2 workgroups), every fragment load misses the 32 KiB vector L1 and the
// per-CU L2->L1 fill path (~9.6 B/clk) co-limits the matrix pipe.  Here the roles are swapped:
//
//   * a wave owns EDGES (2 blocks of 16 = 32 edges) for its whole life and walks ALL row tiles of the conv; the
//     NW = 8 waves of the one workgroup a CU holds walk the same tile sequence at the same time, so a W2 fragment
//     enters the CU's L1 once per 256 edges instead of once per 48;
//   * nothing is staged through LDS on the way to the matrix cores: the radial-MLP input rows are loaded from global
//     memory straight into MFMA B-operand registers (the reduction index is permuted inside every 16-group at pack
//     time, api.cpp), the hidden layer's MFMA result registers ARE the B operand of the W2 tiles -- no transpose, no
//     workgroup barrier, no prologue that a second workgroup would have to hide;
//   * LDS holds only what the contraction reads per tile, wave-private: the gathered x[gth] rows (120 of the 168
//     floats at a time: outputs 0e/1o read inputs 0e,1o,1e; outputs 1e/0o read 0o,1o,1e -- the 48-float scalar slot
//     is swapped once, between the two output halves), the edge harmonics and the l=2 coupling matrix;
//   * the kernel is persistent: one workgroup per CU pulls (conv, 256-edge block, channel part) units from a device
//     queue.  All K=144 convs of an interaction layer (or of the torsion heads) are ONE launch.  Blocks of the last,
//     partial round are split along the output channels into S = 2/4/8 parts (disjoint message columns, no atomics),
//     so that the tail fills the chip with short units instead of leaving CUs idle for a whole block life.
//
// Results are bitwise independent of which workgroup processes a unit and of the split: a message element has one
// owning lane and a fixed summation order (channel-owner accumulation, as in k_conv).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define C2_XLD 124                                   // LDS x row: 120 floats + 4 (odd multiple of 4: 16 rows -> 16 distinct 16-B slots)
#define C2_WAVE_FLOATS (32 * C2_XLD + 32 * 10 + 32 * 8 + 32)   // x rows | harmonics | l=2 matrix | gather indices

// RB: how the waves of the workgroup are kept on the same W2 tiles (the L1 window is ~3 tiles):
//   0  not at all (they drift apart: 41 % of the fragment loads still miss L1, PMC)
//   2  s_barrier at every tile start
//   3  two s_barriers per tile, the second-dispatched half of the waves (the SIMD partners of the first half) running
//      half a tile behind: while one wave of a SIMD contracts, its partner owns the matrix pipe
template <int NW, int RB, int ABL = 0>   // ABL (developer, wrong results): 1 no contraction, 2 no W2 fragment re-load
__global__ __launch_bounds__(64 * NW, 2) void k_conv2(Conv2Args a) {
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
    if (RB == 0 && e0 >= E) continue;                // this wave has no edge in the block (the loop-top barrier is still reached)
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
    // ---- radial-MLP input rows straight into B-operand registers: k-step q of lane group g of 16-group s4 takes
    // k = 16 s4 + 4 g + q (api.cpp pack_conv), i.e. one aligned float4 of the edge's [emb | tab1 | tab2] row
    f32x4 Ba[2][KT];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s4 = 0; s4 < KT; ++s4) {
        const float* src = s4 < 3 ? r0[b] : s4 < 6 ? r1[b] : r2[b];
        Ba[b][s4] = *reinterpret_cast<const f32x4*>(src + 16 * (s4 % 3) + 4 * g);
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
    // ---- hidden layer on the matrix cores, all in registers: h = relu(W1 a + b1); D[row = hidden unit, col = edge] ->
    // lane (g, n) ends up with h[16 m + 4 g + r][edge n], which is exactly the B operand of k-step (s4 = m, q = r)
    const int vW = lane * 16;
    float Bv[2][4 * KT];
    {
      const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.W1p, 0, KT * KT * 1024, 0x00020000);
      f32x4 A1[KT];
#pragma unroll
      for (int s4 = 0; s4 < KT; ++s4) A1[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW1, vW, s4 * 1024, 0));
#pragma unroll
      for (int m = 0; m < KT; ++m) {
        f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
        const int mn = m + 1 < KT ? m + 1 : m;
#pragma unroll
        for (int s4 = 0; s4 < KT; ++s4) {
          const f32x4 av = A1[s4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Ba[0][s4][q], h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Ba[1][s4][q], h1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          A1[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW1, vW, (mn * KT + s4) * 1024, 0));
        }
        const f32x4 b1v = *reinterpret_cast<const f32x4*>(d.w.b1 + 16 * m + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Bv[0][4 * m + r] = fmaxf(h0[r] + b1v[r], 0.f);
          Bv[1][4 * m + r] = fmaxf(h1[r] + b1v[r], 0.f);
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
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.W2q, 0, d.w.n_tiles * KT * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.b2q, 0, d.w.n_tiles * 64, 0x00020000);
    const int vB = g * 16;
    f32x4 A[KT], bias_n = {0.f, 0.f, 0.f, 0.f};
    int t_last = 0;
    if (r_begin < r_end) {
      const RunDesc rl = d.w.runs[r_end - 1];
      t_last = (rl.tile0_n & 0xfffff) + (rl.tile0_n >> 20) - 1;
      const int t0 = d.w.runs[r_begin].tile0_n & 0xfffff;
#pragma unroll
      for (int s4 = 0; s4 < KT; ++s4)
        A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, vW, (t0 * KT + s4) * 1024, 0));
      bias_n = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, vB, t0 * 64, 0));
    }
    float oacc[2][3];
#pragma unroll
    for (int b = 0; b < 2; ++b) oacc[b][0] = oacc[b][1] = oacc[b][2] = 0.f;
    const float* xs_lane = xs + n * C2_XLD;          // + 16 b C2_XLD per edge block
    const float* sh_lane = shs + n * 10;
    if (RB == 3 && wave >= NW / 2) __builtin_amdgcn_s_barrier();   // second half: one barrier interval (half a tile) behind
    // the two waves of a SIMD run the same code from the same start: left alone they stay IN phase (both in their MFMA
    // burst, then both in their contraction, the matrix pipe idle).  The second-dispatched half starts its tiles a
    // fraction of a tile late; the offset then persists (a wave that contracts lets its partner run at full rate)
    if (RB == 0 && wave >= NW / 2) for (int i = 0; i < a.skew; ++i) __builtin_amdgcn_s_sleep(8);
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
          for (int s4 = 0; s4 < KT; ++s4) {
            const f32x4 av = A[s4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Bv[0][4 * s4 + q], acc[0], 0, 0, 0);
              acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Bv[1][4 * s4 + q], acc[1], 0, 0, 0);
            }
            // pin the re-load BEHIND the last MFMA that reads this register (one fragment set, no spills)
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 2)) A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, vW, (tn * KT + s4) * 1024, 0));
            if (RB == 3 && s4 == 3) __builtin_amdgcn_s_barrier();
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
    if (RB == 3 && wave < NW / 2) __builtin_amdgcn_s_barrier();    // first half: waits for the partners' last half tile
  }
  // ---- the last workgroup to leave re-arms the queue for the next launch
  if (tid == 0) {
    const int dn = atomicAdd(a.queue + 1, 1);
    if (dn == n_wg - 1) { a.queue[0] = 0; a.queue[1] = 0; __threadfence(); }
  }
}

void launch_conv2(const Conv2Args& a, hipStream_t st) {
  static int no_split = getenv("DBFR_CONV2_NOSPLIT") ? atoi(getenv("DBFR_CONV2_NOSPLIT")) : 0;
  static int skew = getenv("DBFR_CONV2_SKEW") ? atoi(getenv("DBFR_CONV2_SKEW")) : 0;
  constexpr int NW = 8;
  const size_t lds = (size_t)NW * C2_WAVE_FLOATS * sizeof(float);
  const int n_cu = dbfr_current_cu_count();
  Conv2Args b = a;
  b.skew = skew;
  b.run_barrier = 0;
  b.no_split = no_split;
  // (the LDS attribute is set on every launch: it is per device, and a process may drive several)
#define GO(...) do { if (dbfr_launch_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "k_conv2: hipFuncSetAttribute(MaxDynamicSharedMemorySize)")) return; \
                     hipLaunchKernelGGL((k_conv2<__VA_ARGS__>), dim3(n_cu), dim3(64 * NW), lds, st, b); } while (0)
#ifdef DBFR_DEV_VARIANTS   // developer build (DBFR_BUILD_DEV=1): barrier placements and timing-only ablations (wrong results)
  static int run_barrier = getenv("DBFR_CONV2_BARRIER") ? atoi(getenv("DBFR_CONV2_BARRIER")) : 0;
  static int abl = getenv("DBFR_CONV2_ABL") ? atoi(getenv("DBFR_CONV2_ABL")) : 0;
  b.run_barrier = run_barrier;
  if (abl == 1) { GO(NW, 0, 1); return; }
  if (abl == 2) { GO(NW, 0, 2); return; }
  if (abl == 3) { GO(NW, 0, 3); return; }
  if (run_barrier == 3) { GO(NW, 3); return; }
  if (run_barrier == 2) { GO(NW, 2); return; }
#endif
  GO(NW, 0);
#undef GO
}
