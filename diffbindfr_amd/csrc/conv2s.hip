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
#define SB_CROSS 0x086              // sched_barrier mask: VALU, SALU and LDS instructions may cross (MFMA and VMEM may not)

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
    // One wave issues a v_mfma_f32_16x16x32_bf16 every ~17 cycles whatever its SIMD partner does (two waves of a SIMD
    // run their MFMAs side by side at that rate each: tools/exp/mfma_dep.hip), so matrix time and contraction time of
    // a wave ADD unless the wave itself interleaves them.  The tile is therefore taken block by block,
    //     pass(block 0, tile i)    with   contraction(block 1, tile i-1)   in its issue gaps
    //     pass(block 1, tile i)    with   contraction(block 0, tile i)
    // and the interleaving is written out by hand: the contraction is a list of micro-operations (one LDS read group,
    // then single FMAs), operation k sits behind MFMA k + 4 of the 30 of a pass, and a scheduling barrier after every
    // slot keeps hipcc from re-clustering them (left alone it moves all vector work behind, or in front of, the MFMAs).
    // The W2 re-loads sit behind the last MFMA that reads the registers they overwrite.
    f32x4 acc[2];
    auto pass = [&](auto bc, auto last_c, int tn, auto&& op) {
      constexpr int b = decltype(bc)::value;
      constexpr bool RELOAD = decltype(last_c)::value && !(ABL & 2);   // block 1 is the last reader of the tile's fragments
#define SLOT(m) do { op(std::integral_constant<int, (m) - 4>{}); __builtin_amdgcn_sched_barrier(0); } while (0)
      op(std::integral_constant<int, 0>{});                            // the LDS reads of the travelling contraction
      __builtin_amdgcn_sched_barrier(0);
      acc[b] = bias_n;
      auto step = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const bf16x8 a0 = A[0][s], a1 = A[1][s], a2 = A[2][s];
        // the six partial products with i + j <= 4, smallest first
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, __builtin_bit_cast(bf16x8, Bh[b][0][s]), acc[b], 0, 0, 0); SLOT(6 * s + 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, Bh[b][1][s]), acc[b], 0, 0, 0); SLOT(6 * s + 1);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, Bh[b][2][s]), acc[b], 0, 0, 0); SLOT(6 * s + 2);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, Bh[b][0][s]), acc[b], 0, 0, 0); SLOT(6 * s + 3);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, Bh[b][1][s]), acc[b], 0, 0, 0); SLOT(6 * s + 4);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, __builtin_bit_cast(bf16x8, Bh[b][0][s]), acc[b], 0, 0, 0);
        if (RELOAD) {
          if (s == 0) bias_n = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, vB, tn * 64, 0));
#pragma unroll
          for (int i = 0; i < 3; ++i)
            A[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, vW, tn * C3_TILE_BYTES + (i * 4 + s) * 1024, 0));
        }
        SLOT(6 * s + 5);
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
      // k = 128..143 on v_mfma_f32_16x16x16_bf16 (two-register operands).
      // Hazard (found the hard way, MI355X + ROCm 7.2): an x16 MFMA that takes as SrcC an accumulator an x32 MFMA has just
      // written reads stale data -- hipcc pads nothing between the two opcodes (wrong, timing-dependent results; either
      // opcode alone chains cleanly, tools/exp/mfma_x16_chain.hip).  16 wait states put any pass count behind us.
      asm volatile("s_nop 15");
      __builtin_amdgcn_sched_barrier(0);
      {
        const s16x4 a0 = At[0], a1 = At[1], a2 = At[2];
#define TB(i) __builtin_bit_cast(s16x4, Bt[b][i])
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2, TB(0), acc[b], 0, 0, 0); SLOT(24);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, TB(1), acc[b], 0, 0, 0); SLOT(25);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(2), acc[b], 0, 0, 0); SLOT(26);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, TB(0), acc[b], 0, 0, 0); SLOT(27);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(1), acc[b], 0, 0, 0); SLOT(28);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, TB(0), acc[b], 0, 0, 0);
#undef TB
        if (RELOAD) {
#pragma unroll
          for (int i = 0; i < 3; ++i)
            At[i] = __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(rW, vT, tn * C3_TILE_BYTES + C3_TAIL_OFF + i * 512, 0));
        }
        SLOT(29);
      }
#undef SLOT
    };
    auto no_op = [](auto) {};
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    __builtin_amdgcn_sched_barrier(0);
    if (r_begin < r_end) pass(B0{}, std::false_type{}, 0, no_op);   // block 0 of the part's first tile
    __builtin_amdgcn_sched_barrier(0);
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
        // the contraction of one edge block as micro-operations: K = 0 the LDS reads, K >= 1 one FMA each (K beyond the
        // type's count: nothing).  State lives in cx* / cz* between the operations of one contraction.
        f32x4 cxa, cxb, cxc, cma;
        float2 cmb;
        float cz0, cz1, cz2;
        auto cop = [&](auto bc, auto kc, const float* xp) {
          constexpr int b = decltype(bc)::value;
          constexpr int K = decltype(kc)::value;
          if (ABL & 1) { if (K == 1) asm volatile("" ::"v"(acc[b])); return; }
          const f32x4 v = acc[b];
          const f32x4* x4 = reinterpret_cast<const f32x4*>(xp + 16 * b * C2_XLD);   // [u0..u0+3][3] = 12 consecutive floats (VIN)
          if (VIN && K == 2) cxb = x4[1];            // (each quad requested three operations ahead of its first use: 4 fewer live registers)
          if (VIN && K == 6) cxc = x4[2];
          if (K == 0) {
            cxa = x4[0];
            if (TYPE == PT_VTV) {                      // symmetric traceless matrix of the l=2 harmonics
              cma = *reinterpret_cast<const f32x4*>(ms + (16 * b + n) * 8);
              cmb = *reinterpret_cast<const float2*>(ms + (16 * b + n) * 8 + 4);
            }
          }
          if (!VIN) {
            if (K == 1) cz0 = v[0] * cxa[0];
            if (K == 2) cz0 += v[1] * cxa[1];
            if (K == 3) cz0 += v[2] * cxa[2];
            if (K == 4) cz0 += v[3] * cxa[3];
            if (K == 5) oacc[b][0] += cz0 * S[b][0];
            if (TYPE == PT_SV) {
              if (K == 6) oacc[b][1] += cz0 * S[b][1];
              if (K == 7) oacc[b][2] += cz0 * S[b][2];
            }
          } else {
            if (K == 1) cz0 = v[0] * cxa[0];
            if (K == 2) cz1 = v[0] * cxa[1];
            if (K == 3) cz2 = v[0] * cxa[2];
            if (K == 4) cz0 += v[1] * cxa[3];
            if (K == 5) cz1 += v[1] * cxb[0];
            if (K == 6) cz2 += v[1] * cxb[1];
            if (K == 7) cz0 += v[2] * cxb[2];
            if (K == 8) cz1 += v[2] * cxb[3];
            if (K == 9) cz2 += v[2] * cxc[0];
            if (K == 10) cz0 += v[3] * cxc[1];
            if (K == 11) cz1 += v[3] * cxc[2];
            if (K == 12) cz2 += v[3] * cxc[3];
            if (TYPE == PT_VS) {
              if (K == 13) oacc[b][0] += cz0 * S[b][0];
              if (K == 14) oacc[b][1] += cz1 * S[b][0];
              if (K == 15) oacc[b][2] += cz2 * S[b][0];
            } else if (TYPE == PT_VVS) {
              if (K == 13) oacc[b][0] += cz0 * S[b][0];
              if (K == 14) oacc[b][0] += cz1 * S[b][1];
              if (K == 15) oacc[b][0] += cz2 * S[b][2];
            } else if (TYPE == PT_VVV) {
              if (K == 13) oacc[b][0] += cz1 * S[b][2];
              if (K == 14) oacc[b][1] += cz2 * S[b][0];
              if (K == 15) oacc[b][2] += cz0 * S[b][1];
              if (K == 16) oacc[b][0] -= cz2 * S[b][1];
              if (K == 17) oacc[b][1] -= cz0 * S[b][2];
              if (K == 18) oacc[b][2] -= cz1 * S[b][0];
            } else {   // PT_VTV
              if (K == 13) oacc[b][0] += cma[0] * cz0;
              if (K == 14) oacc[b][1] += cma[1] * cz0;
              if (K == 15) oacc[b][2] += cma[2] * cz0;
              if (K == 16) oacc[b][0] += cma[1] * cz1;
              if (K == 17) oacc[b][1] += cma[3] * cz1;
              if (K == 18) oacc[b][2] += cmb.x * cz1;
              if (K == 19) oacc[b][0] += cma[2] * cz2;
              if (K == 20) oacc[b][1] += cmb.x * cz2;
              if (K == 21) oacc[b][2] += cmb.y * cz2;
            }
          }
        };
        const float* xp = xs_lane + xo;
        for (int i = 0; i < nt; ++i, xp += x_step) {
          const int t = tile0 + i;
          const int tn = t < t_last ? t + 1 : t;
          if (RB >= 2) __builtin_amdgcn_s_barrier();
          pass(B1{}, std::true_type{}, tn, [&](auto kc) { cop(B0{}, kc, xp); });
          // block 0 of the NEXT tile, whichever run it belongs to (behind the part's last tile: computed, never used)
          pass(B0{}, std::false_type{}, 0, [&](auto kc) { cop(B1{}, kc, xp); });
        }
        if (flags & 2) {   // last run of the channel group: this lane owns msg[e][oo .. oo + (VOUT ? 3 : 1))
          if (oo < D_out) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
              if (e0 + 16 * b + n < E) {                       // (edge index recomputed: nothing per edge stays live across the sweep)
                float* op = d.msg + (size_t)(e0 + 16 * b + n) * D_out + oo;
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
#define GO(RB, AB) { if (dbfr_launch_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2s<NW, RB, AB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "k_conv2s: hipFuncSetAttribute(MaxDynamicSharedMemorySize)")) return; \
                     hipLaunchKernelGGL((k_conv2s<NW, RB, AB>), dim3(n_cu), dim3(64 * NW), lds, st, b); return; }
#ifdef DBFR_DEV_VARIANTS   // developer build (DBFR_BUILD_DEV=1): DBFR_CONV2S_VAR = 10 RB + ABL (barrier placement, timing-only ablations)
  static int var = getenv("DBFR_CONV2S_VAR") ? atoi(getenv("DBFR_CONV2S_VAR")) : 0;
  switch (var) {
    case 20: GO(2, 0) case 22: GO(2, 2) case 1: GO(0, 1) case 2: GO(0, 2) case 3: GO(0, 3) case 4: GO(0, 4)
    case 8: GO(0, 8) case 16: GO(0, 16) case 32: GO(0, 32) default: break;
  }
#endif
  GO(0, 0)
#undef GO
}
