// Fused tensor-product convolution, split-fp16 GEMM: the radial MLP's 144 x W GEMM on v_mfma_f32_16x16x32_f16 with every fp32
// operand cut into TWO fp16 pieces -- half the matrix instructions of k_conv2r (conv2r.hip: three bf16 pieces, six products).
//
// Arithmetic.  x = hi + lo + d with hi = fp16(x), lo = fp16(x - hi), both rounded to nearest even: 2 x 11 significand bits plus
// the sign of lo leave |d| <= 2^-23 |x|.  A product w h is evaluated as hi_w lo_h + lo_w hi_h + hi_w hi_h (the dropped lo_w lo_h
// is below 2^-22 |w h|), each term exact in the fp32 accumulator.  fp16 has five exponent bits, so both operands are first
// multiplied by exact powers of two that put their largest magnitude into [2^14, 2^15):
//   * W2: per RUN (the tiles of one tensor-product path of one channel group), at model creation (api.cpp pack_conv2); the
//     bias rows carry the same factor 2^k, and 2^-k is folded into the run's harmonics (l = 2 matrix), i.e. it costs nothing
//     per tile;
//   * h = relu(W1 a + b1): per EDGE, in the kernel (largest of the edge's 144 activations); the bias is multiplied by the
//     edge's factor where k_conv2r copied it into the accumulator, and the inverse goes into the message store.
// The small products of all five k-steps (4 x 32 k + the last 16 k, whose two small products share ONE x32 MFMA: A = [hi_w | lo_w],
// B = [lo_h | hi_h]) and the five large ones run in SEPARATE accumulators (small ones start from the bias, large ones from 0) that are
// added once per tile -- where k_conv2r moved its accumulator into the previous-tile
// registers: the large chain is rounded 5 times instead of 15, which is what puts the error below the fp32 instruction's
// (tools/exp/split_f16.hip, profiles/r3_split_experiments.txt).
//
// Structure: k_conv2r's (persistent edge-owner waves, unit queue, tail split, wave-private x rows / harmonics, channel-owner
// accumulation, previous tile's contraction hand-interleaved into the MFMA issue gaps), with a leaner LDS ring because a
// tile is now 9 KiB of pieces and half as many MFMAs long:
//   * TWO barriers per tile instead of three, the minimum for a one-tile ring (every slot is written once and read once per
//     tile, and both orders need a barrier between them).  Slots of k-steps 0..2 ("A") are read during k-steps 4, 0, 1 and
//     written during k-step 3 (window: 2, 3); slots of k-step 3 and of the last 16 k ("B") are read during k-steps 2, 3 and
//     written during k-step 1 (window: 4, 0, 1); barriers open k-steps 2 and 4;
//   * a wave moves ONE 1-KiB share of A (16 bytes per lane) and ONE 512-byte share of B per tile -- 2 loads + 2 LDS writes per
//     tile instead of 6 + 6.  A has seven shares: the six 1-KiB pieces and the tile's 16 bias values (64 bytes; the other
//     lanes of that share are out of the buffer's range, get zeros and write them into padding); waves 6, 7 both move the bias
//     share, and waves 6, 7 duplicate B shares 0, 1: same bytes to the same place, no branch;
//   * a share is re-fetched right behind its write, i.e. it travels for almost a whole tile (~ 0.8 us) before it is needed.  (Tried:
//     two staging sets by tile parity = a distance of two tiles.  The uniform branch between the two tile bodies makes hipcc's
//     wait-count pass wait for vmcnt(0) in front of every write, and with the staging registers taken out of the compiler's hands --
//     v[244:255] behind amdgpu_num_vgpr(244), loads / waits / writes in inline assembly -- the kernel was correct but 1-2 % SLOWER
//     than this form, and the reservation turned out not to be binding: another instantiation of the same template used v244 / v245
//     as temporaries.  The W2 pieces are L2 hits 92 % of the time, profiles/r3_pmc_k_conv2h.json.)
// The x32 -> x16 accumulator hazard of conv2r.hip does not arise: with two accumulator pairs the one x16 MFMA per block takes as
// SrcC an accumulator whose last writer is at least two MFMAs (>= 32 issue cycles) back.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CH_TILE_BYTES 9280          // [2 pieces][4 k-steps of 32][64 lanes][8 fp16] + [2 pieces][64 lanes][4 fp16] (the last 16 k) + 16 bias values
#define CH_TAIL_OFF 8192
#define CH_BIAS_OFF 9216
#define CH_RING_BYTES 10240         // one tile; the bias share is 1 KiB wide in LDS (64 bytes of bias + the zeros of its idle lanes)

// two fp32 values -> hi and lo words of packed fp16 pairs (low half = x0's piece), round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ void split2x2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector((f32x2){x0, x1}, f16x2);
  const f16x2 l = __builtin_convertvector((f32x2){x0 - (float)h[0], x1 - (float)h[1]}, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

// element by element on purpose: as vector operations hipcc emits v_pk_mul_f32 / v_pk_add_f32, which cost more next to MFMAs than
// the two plain instructions they replace (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
__device__ __forceinline__ void scale4(f32x4& o, const f32x4& v, float s) { o[0] = v[0] * s; o[1] = v[1] * s; o[2] = v[2] * s; o[3] = v[3] * s; }
__device__ __forceinline__ void sum4(f32x4& o, const f32x4& a, const f32x4& b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = a[3] + b[3]; }

#define C2_XLD 124                                   // LDS x row: 120 floats + 4 (odd multiple of 4: 16 rows -> 16 distinct 16-B slots)
#define C2_WAVE_FLOATS (32 * C2_XLD + 32 * 10 + 32 * 8 + 32)   // x rows | harmonics | l=2 matrix | gather indices

// ABL & 1024 (a PRODUCT variant, right results): the rows of W2h carry their own powers of two (ConvW2::W2rinv: convs whose rows lie further apart
// than two fp16 pieces hold behind one factor per run); their inverses come off the accumulator rows where the tile's two chains are added.
template <int NW, int ABL = 0>   // ABL (developer, wrong results): 1 no contraction, 2 no ring barriers, 4 no ring filling (8: loads only, 16: LDS writes only), 32 no per-tile fetch (the unit's first shares are written again and again: real data, no loads), 512 every fetch from tiles 0..7 (always L2 hits), 64 unit prologue only; 256 (right results): ring writes right behind the barriers (slots 0 / 12 instead of 9 / 21)
__global__ __launch_bounds__(64 * NW, 2) void k_conv2h(Conv2Args a) {
  constexpr int K = 144, KT = 9;
  constexpr bool ROWF = (ABL & 1024) != 0;
  constexpr int EPB = 32 * NW;                       // edges per block (unit)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  // Static priority for the second-dispatched half of the workgroup (waves w and w + 4 share a SIMD; the younger one loses every
  // arbitration otherwise: MI355X_MICROARCH.md, "two waves per SIMD", item 4).  Same-box A/B, two rounds: 3.488 -> 3.464 ms per conv,
  // 313.4 -> 314.7 poses/s (profiles/r4_ab_prio.txt).
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  const char* ring = reinterpret_cast<const char*>(lds);          // [CH_RING_BYTES] first: inside the 16-bit offset of ds_read
  float* xs = lds + CH_RING_BYTES / 4 + wave * C2_WAVE_FLOATS;    // [32][C2_XLD]
  float* shs = xs + 32 * C2_XLD;                     // [32][10]
  float* ms = shs + 32 * 10;                         // [32][8]
  int* s_gth = reinterpret_cast<int*>(ms + 32 * 8);  // [32]

  // developer (DBFR_CONV2_TRACE=<file>): shader clock / 100-MHz wall clock at the start and the end of workgroup 0 -> effective clock
  if (a.trace && blockIdx.x == 0 && tid == 0) { a.trace[0] = __builtin_readcyclecounter(); a.trace[1] = __builtin_amdgcn_s_memrealtime(); }
  // ---- the launch's unit list (every workgroup derives it from the device-side edge counts)
  int nb0 = 0, nb1 = 0, nb2 = 0, nb3 = 0;
  {
    auto blocks = [&](int c) { return c < a.n_conv ? (min(*a.c[c].n_edges, a.c[c].max_edges) + EPB - 1) / EPB : 0; };
    nb0 = blocks(0); nb1 = blocks(1); nb2 = blocks(2); nb3 = blocks(3);
  }
  const int N = nb0 + nb1 + nb2 + nb3;
  const int n_wg = gridDim.x;
  const int full = (N / n_wg) * n_wg, rem = N - full;
  // The blocks of the last, partial round are cut 2 / 4 / 8-way along the output channels (disjoint message columns).  Which cut: the one
  // with the smallest (rounds of parts) x (part length), a part costing 1 / 2^si of a block + ~5 % for its own prologue -- e.g. 140 blocks
  // on 256 workgroups: uncut 1 x 1.05, 2-way 2 x 0.55, 4-way 3 x 0.30, 8-way 5 x 0.17 = 0.86 (the old rule "cut only if every part finds
  // a free workgroup" left 116 of 256 CUs idle for such a launch: predict.py-sized batches).  A message does not depend on the cut.
  int si = 0;
  if (rem > 0 && !a.no_split) {
    int best = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cost = (((rem << c) + n_wg - 1) / n_wg) * ((128 >> c) + 6);
      if (cost < best) { best = cost; si = c; }
    }
  }
  const int total = full + (rem << si);

  // Units are dealt round robin: workgroup b takes units b, b + n_wg, ...  All units of a launch cost the same (256 edges x the tile count
  // of their conv / part; the four convs of a layer share one W), so a dynamic queue balances nothing here, and its returning atomic --
  // 1-3 us with 256 workgroups pulling, in front of a barrier all eight waves wait at -- was 0.5-1 % of a 240-us unit (round 4).
  for (int it = 0;; ++it) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // nothing of the last unit still reads or arms the ring
    __syncthreads();
    const int u = (int)blockIdx.x + it * n_wg;
    if (u >= total) break;
    int blk, part = 0, psi = 0;
    if (u < full) blk = u;
    else { const int v = u - full; blk = full + (v >> si); part = v & ((1 << si) - 1); psi = si; }
    int c = 0;
    if (blk >= nb0) { blk -= nb0; c = 1; if (blk >= nb1) { blk -= nb1; c = 2; if (blk >= nb2) { blk -= nb2; c = 3; } } }
    const Conv2Desc& d = a.c[c];
    const int E = min(*d.n_edges, d.max_edges);
    const int e0 = blk * EPB + 32 * wave;
    // (a wave without edges in this block runs along on clamped edges and stores nothing: the ring barriers need all eight)
    const int D_in = d.w.D_in, D_out = d.w.D_out;
    const int vW = lane * 16, vB = g * 16;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int r_begin = d.w.part_run[psi][part], r_end = d.w.part_run[psi][part + 1];
    int t_last = 0, t_first = 0;
    if (r_begin < r_end) {
      const RunDesc rl = d.w.runs[r_end - 1];
      t_last = (rl.tile0_n & 0xfffff) + (rl.tile0_n >> 20) - 1;
      t_first = d.w.runs[r_begin].tile0_n & 0xfffff;
    }
    // Filling the ring (header comment).  Share A = one of the six 1-KiB pieces (piece i, k-step s < 3) or the bias values; share B =
    // one of the six 512-byte halves of k-step 3's two pieces and of the two last-16-k pieces.  Same layout in memory and in LDS.
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.W2h, 0, d.w.n_tiles * CH_TILE_BYTES, 0x00020000);
    const int shA = wv & 7, shB = wv % 6;
    const int offA = shA < 6 ? ((shA / 3) * 4 + shA % 3) * 1024 : CH_BIAS_OFF;
    const int offB = shB < 4 ? ((shB >> 1) * 4 + 3) * 1024 + (shB & 1) * 512 : CH_TAIL_OFF + (shB - 4) * 512;
    const int vA = (shA < 6 || lane < 4) ? lane * 16 : 0x40000000;      // lanes 4..63 of the bias share: out of range, the load returns 0
    char* ringw = const_cast<char*>(ring);
    u32x4 stgA = {0u, 0u, 0u, 0u};                     // staging registers: a share is fetched right behind the write of its predecessor
    u32x2 stgB = {0u, 0u};
    auto fetchA = [&](int tile) { if (!(ABL & (4 | 16))) stgA = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, vA, ((ABL & 512) ? (tile & 7) : tile) * CH_TILE_BYTES + offA, 0)); };
    auto fetchB = [&](int tile) { if (!(ABL & (4 | 16))) stgB = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rW, lane * 8, ((ABL & 512) ? (tile & 7) : tile) * CH_TILE_BYTES + offB, 0)); };
    auto putA = [&] { if (!(ABL & (4 | 8))) *reinterpret_cast<u32x4*>(ringw + offA + vW) = stgA; };
    auto putB = [&] { if (!(ABL & (4 | 8))) *reinterpret_cast<u32x2*>(ringw + offB + lane * 8) = stgB; };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    if (r_begin < r_end) {   // the part's first tile travels while the hidden layer is computed
      fetchA(t_first); fetchB(t_first);
    }

    // ---- my two edges (block b, column n), clamped; gather indices
    int ev[2], gthv[2];
    const float* r0[2]; const float* r1[2]; const float* r2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int e = e0 + 16 * b + n;
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
    // ---- hidden layer h = relu(W1 a + b1) on the fp16 matrix instruction too (same two-piece / three-product / two-accumulator
    // form as the W2 tiles below; on v_mfma_f32_16x16x4_f32 its 648 MFMAs of 32 cycles were 7 % of the kernel at W = 7776 and 17 %
    // at W = 2880).  The radial-MLP input row a = [emb | tab1 | tab2] of the lane's edge goes straight into B-operand registers
    // (k-step q of lane group g of 16-group s4 takes k = 16 s4 + 4 g + q, api.cpp pack_conv): scaled per edge by the power of
    // two that puts max |a| into [2^14, 2^15) and cut into fp16 pieces.  W1 comes as nine 16-row tiles in the W2h tile format
    // (api.cpp: W1h, ONE factor 2^k1 for the matrix, bias rows x 2^k1), fetched per wave through L1 (83 KB, L2 resident), both
    // edge blocks per fragment.  D[row = hidden unit, col = edge] -> lane (g, n) ends up with H' = relu(acc) = h 2^(k1 + ja) for
    // hidden units 16 m + 4 g + r, r = 0..3; no need to take the factor off: the edge's 144 values (36 in each of the four lanes
    // n, n + 16, n + 32, n + 48) are scaled once more so that their maximum lies in [2^14, 2^15), cut into fp16 pieces and filed
    // as slots 4 (m & 1) + r of k-step m >> 1 of the W2 tiles' B operand; se / ue carry the total factor and its inverse.
    u32x4 Bh[2][2][4];                                 // h pieces: [edge block][hi, lo][k-step of 32] = 8 fp16 each, 64 VGPRs
    u32x4 Btc[2];                                      // ... and of the last 16 k: [lo (4 fp16) | hi (4 fp16)] -- the B operand of the ONE x32 MFMA that
                                                       // carries both small products of these 16 k; its upper half is the x16 MFMA's B operand
    float se[2], ue[2];                                // the edge's factor 2^j on h and its inverse
    {
      const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.W1h, 0, KT * CH_TILE_BYTES, 0x00020000);
      u32x4 Ah[2][2][4];                               // a pieces, same filing as Bh
      u32x4 Atc[2];                                    // ... of the last 16 k: [lo | hi] (see Btc)
      float sa[2];
      int ja[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f32x4 Ba[KT];
        float amx = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < KT; ++s4) {
          const float* src = s4 < 3 ? r0[b] : s4 < 6 ? r1[b] : r2[b];
          Ba[s4] = *reinterpret_cast<const f32x4*>(src + 16 * (s4 % 3) + 4 * g);
#pragma unroll
          for (int q = 0; q < 4; ++q) amx = fmaxf(amx, fabsf(Ba[s4][q]));
        }
        amx = fmaxf(amx, __shfl_xor(amx, 16));
        amx = fmaxf(amx, __shfl_xor(amx, 32));
        // (inputs below 2^-14 are not blown up further: the bias is of order 1.  No clamp on the large side: sa = 2^(15 - ja) is a normal
        // fp32 number for every finite input, and a clamp there would let a sa overflow fp16 for inputs beyond it)
        ja[b] = max(-14, __builtin_amdgcn_frexp_expf(amx));
        sa[b] = __builtin_amdgcn_ldexpf(1.f, 15 - ja[b]);
#pragma unroll
        for (int s4 = 0; s4 < KT; ++s4) {
          unsigned hi0, lo0, hi1, lo1;
          split2x2(Ba[s4][0] * sa[b], Ba[s4][1] * sa[b], hi0, lo0);
          split2x2(Ba[s4][2] * sa[b], Ba[s4][3] * sa[b], hi1, lo1);
          if (s4 < 8) {
            Ah[b][0][s4 >> 1][2 * (s4 & 1)] = hi0; Ah[b][0][s4 >> 1][2 * (s4 & 1) + 1] = hi1;
            Ah[b][1][s4 >> 1][2 * (s4 & 1)] = lo0; Ah[b][1][s4 >> 1][2 * (s4 & 1) + 1] = lo1;
          } else {
            Atc[b] = (u32x4){lo0, lo1, hi0, hi1};
          }
        }
      }
      // 9 row tiles x (4 k-steps of 32 + the last 16 k); fragments of k-step i + 2 requested behind the MFMAs of k-step i
      constexpr int NST = KT * 4;                      // x32 steps, numbered m * 4 + s
      u32x4 F[3][2];                                   // ring of three k-steps of (hi, lo) fragments
      u32x4 Ft;                                        // the current tile's last-16-k fragments [hi | lo]
      f32x4 bb;                                        // ... and its bias values x 2^k1
      auto ld_step = [&](int i, u32x4 (&f)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) f[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rW1, vW, (i >> 2) * CH_TILE_BYTES + (p * 4 + (i & 3)) * 1024, 0));
      };
      auto ld_tail = [&](int m) {
        Ft = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rW1, vW, m * CH_TILE_BYTES + CH_TAIL_OFF, 0));
        bb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW1, vB, m * CH_TILE_BYTES + CH_BIAS_OFF, 0));
      };
      ld_step(0, F[0]); ld_step(1, F[1]); ld_tail(0);
      float H[2][KT][4];
      float mx[2] = {0.f, 0.f};
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#define AHB(b, p, s) __builtin_bit_cast(f16x8, Ah[b][p][s])
#define FW(i, p) __builtin_bit_cast(f16x8, F[(i) % 3][p])
#pragma unroll
      for (int m = 0; m < KT; ++m) {
        f32x4 aS[2], aB[2];
        scale4(aS[0], bb, sa[0]); scale4(aS[1], bb, sa[1]);
        aB[0] = zero; aB[1] = zero;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int i = m * 4 + s;
          aS[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW(i, 0), AHB(0, 1, s), aS[0], 0, 0, 0);
          aS[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW(i, 0), AHB(1, 1, s), aS[1], 0, 0, 0);
          aS[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW(i, 1), AHB(0, 0, s), aS[0], 0, 0, 0);
          aS[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW(i, 1), AHB(1, 0, s), aS[1], 0, 0, 0);
          aB[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW(i, 0), AHB(0, 0, s), aB[0], 0, 0, 0);
          aB[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW(i, 0), AHB(1, 0, s), aB[1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (i + 2 < NST) ld_step(i + 2, F[(i + 2) % 3]);
        }
        {   // k = 128..143: both small products in ONE x32 MFMA (A = [hi_w | lo_w], B = [lo_a | hi_a]: the 32 k slots of the instruction
            // are the 16 k's twice), the large one on the x16 instruction (its SrcC was last written three MFMAs ago: no x32 -> x16 hazard)
          const f16x8 fw = __builtin_bit_cast(f16x8, Ft);
          const f16x4 fh = __builtin_bit_cast(f16x4, (u32x2){Ft[0], Ft[1]});
          aS[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw, __builtin_bit_cast(f16x8, Atc[0]), aS[0], 0, 0, 0);
          aS[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw, __builtin_bit_cast(f16x8, Atc[1]), aS[1], 0, 0, 0);
          aB[0] = __builtin_amdgcn_mfma_f32_16x16x16f16(fh, __builtin_bit_cast(f16x4, (u32x2){Atc[0][2], Atc[0][3]}), aB[0], 0, 0, 0);
          aB[1] = __builtin_amdgcn_mfma_f32_16x16x16f16(fh, __builtin_bit_cast(f16x4, (u32x2){Atc[1][2], Atc[1][3]}), aB[1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (m + 1 < KT) ld_tail(m + 1);
        }
        f32x4 rv = {1.f, 1.f, 1.f, 1.f};                  // ROWF: W1h's per-row factors (a conv with hidden-layer rows far apart) come off the units here
        if constexpr (ROWF) if (d.w.W1rinv) rv = *reinterpret_cast<const f32x4*>(d.w.W1rinv + 16 * m + 4 * g);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) { H[b][m][r] = fmaxf(aS[b][r] + aB[b][r], 0.f); if constexpr (ROWF) H[b][m][r] *= rv[r]; mx[b] = fmaxf(mx[b], H[b][m][r]); }
      }
#undef AHB
#undef FW
      const int k1 = d.w.k1;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float m2 = fmaxf(mx[b], __shfl_xor(mx[b], 16));
        m2 = fmaxf(m2, __shfl_xor(m2, 32));
        // H' = f 2^j, f in [0.5, 1)  ->  factor 2^(15 - j) on H' = 2^tot on h with tot = 15 - j + k1 + 15 - ja.  tot is what multiplies
        // the W2 bias (already x 2^k of its run) in the accumulator: it is kept below 2^64 so that the product stays a finite fp32
        // number whatever the activations (an edge whose largest activation is below 2^-49 gets a smaller factor than the window
        // asks for: its pieces lose relative precision, next to an O(1) bias nothing to lose), and the factor on H' follows from it.
        // On the other side only fp32's own range limits it (se, ue = 2^+-tot stay normal numbers): huge activations keep their window --
        // a tighter clamp there would push H' t beyond fp16's largest number -- and the bias term, 2^-24 of the products by then,
        // may flush to zero.
        const int j = __builtin_amdgcn_frexp_expf(m2);
        const int tot = max(-120, min(64, 15 - j + k1 + 15 - ja[b]));
        const float t = __builtin_amdgcn_ldexpf(1.f, tot - k1 - 15 + ja[b]);
        se[b] = __builtin_amdgcn_ldexpf(1.f, tot);
        ue[b] = __builtin_amdgcn_ldexpf(1.f, -tot);
#pragma unroll
        for (int m = 0; m < KT; ++m) {
          unsigned hi0, lo0, hi1, lo1;
          split2x2(H[b][m][0] * t, H[b][m][1] * t, hi0, lo0);
          split2x2(H[b][m][2] * t, H[b][m][3] * t, hi1, lo1);
          if (m < 8) {
            Bh[b][0][m >> 1][2 * (m & 1)] = hi0; Bh[b][0][m >> 1][2 * (m & 1) + 1] = hi1;
            Bh[b][1][m >> 1][2 * (m & 1)] = lo0; Bh[b][1][m >> 1][2 * (m & 1) + 1] = lo1;
          } else {
            Btc[b] = (u32x4){lo0, lo1, hi0, hi1};
          }
        }
      }
    }
    // ---- x[gth] rows of the wave's 32 edges into its LDS region (columns [0, min(D_in, 120)))
    __builtin_amdgcn_wave_barrier();
    {   // lane L moves the 16-byte chunks (L >> 5), (L >> 5) + 2, ... of row L & 31: all loads out before the first store, no
        // index arithmetic.  Chunks past the row's end repeat its last one: same bytes to the same place.
      const int d4 = min(D_in, 120) >> 2;
      const float* xsrc = d.x + (size_t)s_gth[lane & 31] * d.ldx;
      float* xdst = xs + (lane & 31) * C2_XLD;
      f32x4 v[15];
#pragma unroll
      for (int j = 0; j < 15; ++j) v[j] = *reinterpret_cast<const f32x4*>(xsrc + 4 * min(2 * j + (lane >> 5), d4 - 1));
#pragma unroll
      for (int j = 0; j < 15; ++j) *reinterpret_cast<f32x4*>(xdst + 4 * min(2 * j + (lane >> 5), d4 - 1)) = v[j];
    }
    __builtin_amdgcn_wave_barrier();

    if (ABL & 64) {   // developer: the unit prologue alone
      asm volatile("" ::"v"(Bh[0][0][0]), "v"(Bh[1][1][3]), "v"(Btc[1]));
      continue;
    }
    // ---- the W2 row tiles of this part, run by run (channel-owner order, api.cpp pack_conv2)
    float oacc[2][3];
#pragma unroll
    for (int b = 0; b < 2; ++b) oacc[b][0] = oacc[b][1] = oacc[b][2] = 0.f;
    const float* xs_lane = xs + n * C2_XLD;          // + 16 b C2_XLD per edge block
    const float* sh_lane = shs + n * 10;
    f16x8 FA[2], FB[2];                                // W2 pieces (hi, lo) of two k-steps: 16 VGPRs
    f16x8 FT;                                          // ... and of the last 16 k [hi | lo]: 4
    f32x4 bias_n;                                      // the NEXT tile's four bias values of this lane's rows (x 2^k of its run)
    f32x4 accN[2];                                     // bias x the edge's factor: where the next tile's small-product chain starts
    f32x4 accS[2], accB[2];                            // this tile's accumulators: small products (+ bias) | large products
    f32x4 accp[2];                                     // the previous tile's result (accS + accB), being contracted
    f32x4 rinv_c = {1.f, 1.f, 1.f, 1.f}, rinv_n = {1.f, 1.f, 1.f, 1.f};   // ROWF: 2^-d of this lane's four rows, of this tile | of the next one
    const float* rinv_base = ROWF && d.w.W2rinv ? d.w.W2rinv + 4 * g : nullptr;
    auto sumr4 = [&](f32x4& o, const f32x4& x, const f32x4& y) {   // the tile's result: small-product chain + large-product chain (x the rows' factors)
      if constexpr (ROWF) { o[0] = (x[0] + y[0]) * rinv_c[0]; o[1] = (x[1] + y[1]) * rinv_c[1]; o[2] = (x[2] + y[2]) * rinv_c[2]; o[3] = (x[3] + y[3]) * rinv_c[3]; }
      else sum4(o, x, y);
    };
    auto rd_step = [&](auto jc, f16x8 (&F)[2]) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int i = 0; i < 2; ++i) F[i] = *reinterpret_cast<const f16x8*>(ring + (i * 4 + j) * 1024 + vW);
    };
    auto ring_barrier = [&] {   // "my LDS reads and writes are done" + s_barrier
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
#define HB(b, p, s) __builtin_bit_cast(f16x8, Bh[b][p][s])
#define M32(C, wp, b, hp, F, s) __builtin_amdgcn_mfma_f32_16x16x32_f16(F[wp], HB(b, hp, s), C, 0, 0, 0)
// the last 16 k: both small products in one x32 MFMA (A = [hi_w | lo_w], B = [lo_h | hi_h]), the large one on the x16 instruction
#define MT32(C, b) __builtin_amdgcn_mfma_f32_16x16x32_f16(FT, __builtin_bit_cast(f16x8, Btc[b]), C, 0, 0, 0)
#define MT16(C, b) __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(FT, FT, 0, 1, 2, 3), __builtin_bit_cast(f16x4, (u32x2){Btc[b][2], Btc[b][3]}), C, 0, 0, 0)
#define SLOT(m) do { op(std::integral_constant<int, (m)>{}); __builtin_amdgcn_sched_barrier(0); } while (0)
    // one tile: 28 MFMAs = 4 k-steps of 32 x (hi_w lo_h, lo_w hi_h | hi_w hi_h) + the last 16 k (both small products in one x32 MFMA | the
    // large one on the x16 instruction), x 2 edge blocks; `op(m)` = what travels behind MFMA m.
    // e0 / e1 / e2: the k-step's memory operations, behind its MFMAs 0 / 1 / 2
    // Tile t writes the staged shares -- its own B shares, the next tile's A shares -- late in their windows (k-steps 1 and 3: right
    // behind a barrier all eight waves would write at once, in front of everybody's fragment reads: -3 %) and re-arms the registers.
    auto tile = [&](int t, auto&& op) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      // k-step 0 (fragments FA): B shares of THIS tile into the ring, next tile's bias on its way, k-step 1's fragments requested
      accS[0] = M32(accN[0], 0, 0, 1, FA, 0); if (ABL & 256) putB(); SLOT(0);
      accS[1] = M32(accN[1], 0, 1, 1, FA, 0); if (ABL & 256) fetchB(min(t + 1, t_last)); SLOT(1);
      accS[0] = M32(accS[0], 1, 0, 0, FA, 0); rd_step(I1{}, FB); SLOT(2);
      accS[1] = M32(accS[1], 1, 1, 0, FA, 0); SLOT(3);
      accB[0] = M32(zero, 0, 0, 0, FA, 0); SLOT(4);
      accB[1] = M32(zero, 0, 1, 0, FA, 0); SLOT(5);
      // k-step 1 (FB): the next tile's B shares set out
      accS[0] = M32(accS[0], 0, 0, 1, FB, 1); SLOT(6);
      accS[1] = M32(accS[1], 0, 1, 1, FB, 1); SLOT(7);
      accS[0] = M32(accS[0], 1, 0, 0, FB, 1); rd_step(I2{}, FA); SLOT(8);
      accS[1] = M32(accS[1], 1, 1, 0, FB, 1); if (!(ABL & 256)) putB(); SLOT(9);
      accB[0] = M32(accB[0], 0, 0, 0, FB, 1); if (!(ABL & (256 | 32))) fetchB(min(t + 1, t_last)); SLOT(10);
      accB[1] = M32(accB[1], 0, 1, 0, FB, 1); SLOT(11);
      // k-step 2 (FA): everybody has read slots 0..2 -> the next tile's A shares go in
      ring_barrier();
      accS[0] = M32(accS[0], 0, 0, 1, FA, 2); if (ABL & 256) putA(); SLOT(12);
      accS[1] = M32(accS[1], 0, 1, 1, FA, 2); if (ABL & 256) fetchA(min(t + 2, t_last)); SLOT(13);
      accS[0] = M32(accS[0], 1, 0, 0, FA, 2); rd_step(I3{}, FB); SLOT(14);
      accS[1] = M32(accS[1], 1, 1, 0, FA, 2); SLOT(15);
      accB[0] = M32(accB[0], 0, 0, 0, FA, 2); SLOT(16);
      accB[1] = M32(accB[1], 0, 1, 0, FA, 2); SLOT(17);
      // k-step 3 (FB): the A shares of the tile after next set out
      accS[0] = M32(accS[0], 0, 0, 1, FB, 3); SLOT(18);
      accS[1] = M32(accS[1], 0, 1, 1, FB, 3); SLOT(19);
      accS[0] = M32(accS[0], 1, 0, 0, FB, 3); FT = *reinterpret_cast<const f16x8*>(ring + CH_TAIL_OFF + vW);
      if constexpr (ROWF) if (rinv_base) rinv_n = *reinterpret_cast<const f32x4*>(rinv_base + 16 * min(t + 1, t_last));   // (a tile ahead, straight from L2: 64 bytes per tile)
      SLOT(20);
      accS[1] = M32(accS[1], 1, 1, 0, FB, 3); if (!(ABL & 256)) putA(); SLOT(21);
      accB[0] = M32(accB[0], 0, 0, 0, FB, 3); if (!(ABL & (256 | 32))) fetchA(min(t + 2, t_last)); SLOT(22);
      accB[1] = M32(accB[1], 0, 1, 0, FB, 3); SLOT(23);
      // k = 128..143: one x32 MFMA (both small products) + one x16 MFMA (the large one) per block; the next tile's slots 0..2 are
      // complete: its first k-step's fragments.  Block 0 finishes two MFMAs before the tile does, so its sum (slot 27) does not wait
      // for the pipe.  (x16 SrcC last written by an x32 MFMA three MFMAs earlier: no hazard, header comment.)
      ring_barrier();
      accS[0] = MT32(accS[0], 0); bias_n = *reinterpret_cast<const f32x4*>(ring + CH_BIAS_OFF + vB); SLOT(24);
      accB[0] = MT16(accB[0], 0); rd_step(I0{}, FA); SLOT(25);
      accS[1] = MT32(accS[1], 1); SLOT(26);
      accB[1] = MT16(accB[1], 1); SLOT(27);
      sumr4(accp[1], accS[1], accB[1]);
      if constexpr (ROWF) rinv_c = rinv_n;
      __builtin_amdgcn_sched_barrier(0);
    };
#undef SLOT
    int x_phase = 0;
    if (r_begin < r_end) {                             // the part's first tile: A shares (+ bias) into the ring, its B shares wait in their
      putA();                                          // staging registers for the first k-step 1; the second tile's A shares set out
      fetchA(min(t_first + 1, t_last));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      rd_step(I0{}, FA);
      bias_n = *reinterpret_cast<const f32x4*>(ring + CH_BIAS_OFF + vB);
      scale4(accN[0], bias_n, se[0]); scale4(accN[1], bias_n, se[1]);
      if constexpr (ROWF) if (rinv_base) rinv_c = *reinterpret_cast<const f32x4*>(rinv_base + 16 * t_first);
    }
    for (int r = r_begin; r < r_end; ++r) {
      const RunDesc rd = d.w.runs[r];
      const int tile0 = rd.tile0_n & 0xfffff, nt = rd.tile0_n >> 20;
      const int type = rd.meta & 15, flags = (rd.meta >> 4) & 3, sh_off = (rd.meta >> 8) & 15, x_step = (rd.meta >> 12) & 0xff;
      const int xph = (rd.meta >> 20) & 1;
      const float run_inv = __builtin_amdgcn_ldexpf(1.f, -((int)rd.meta >> 24));   // the run's W2h / b2h rows carry 2^k
      const int xo = (rd.x_off4 >> (8 * g)) & 0xff, oo = (rd.o_off4 >> (8 * g)) & 0xff;
      if (xph != x_phase) {   // second output half: the scalar slot [0,48) now holds the 48x0o inputs x[120:168]
        __builtin_amdgcn_wave_barrier();
        {
          const float* xsrc = d.x + (size_t)s_gth[lane & 31] * d.ldx + 120;
          float* xdst = xs + (lane & 31) * C2_XLD;
          f32x4 v[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) v[j] = *reinterpret_cast<const f32x4*>(xsrc + 4 * (2 * j + (lane >> 5)));
#pragma unroll
          for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(xdst + 4 * (2 * j + (lane >> 5))) = v[j];
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
        // the run's harmonics (PT_VTV: the symmetric traceless matrix of the l = 2 ones) x 2^-k, once per run
        float S[2][NSV ? NSV : 1];
        float Mv[2][TYPE == PT_VTV ? 6 : 1];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (TYPE != PT_VTV) {
            const float* sp = sh_lane + 160 * b + sh_off;
#pragma unroll
            for (int k = 0; k < NSV; ++k) S[b][k] = sp[k] * run_inv;
          } else {
            const float* mp = ms + (16 * b + n) * 8;
#pragma unroll
            for (int k = 0; k < 6; ++k) Mv[b][k] = mp[k] * run_inv;
          }
        }
        // the contraction of one edge block of the PREVIOUS tile (accp) as micro-operations: K = 0 the first LDS read, K >= 1
        // one FMA each (K beyond the type's count: nothing).  The coupling with the harmonics is linear in cz and the harmonics are
        // constant over a run, so a tile only ACCUMULATES cz = sum_u w[u] x[u] (4 FMAs for scalar inputs, 12 for vector inputs) and the
        // harmonics are applied once per run (`finish`): 2 .. 18 vector instructions per tile less than doing it tile by tile, and
        // fewer roundings.  State lives in cx* / czr between the operations.
        f32x4 cxa, cxb, cxc;
        float czr[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        auto cop = [&](auto bc, auto kc, const float* xp) {
          constexpr int b = decltype(bc)::value;
          constexpr int K = decltype(kc)::value;
          if (ABL & 1) { if (K == 1) asm volatile("" ::"v"(accp[b])); return; }
          const f32x4 v = accp[b];
          const f32x4* x4 = reinterpret_cast<const f32x4*>(xp + 16 * b * C2_XLD);   // [u0..u0+3][3] = 12 consecutive floats (VIN)
          if (VIN && K == 2) cxb = x4[1];            // (each quad requested a few operations ahead of its first use)
          if (VIN && K == 6) cxc = x4[2];
          if (K == 0) cxa = x4[0];
          if (!VIN) {
            if (K == 1) czr[b][0] += v[0] * cxa[0];
            if (K == 2) czr[b][0] += v[1] * cxa[1];
            if (K == 3) czr[b][0] += v[2] * cxa[2];
            if (K == 4) czr[b][0] += v[3] * cxa[3];
          } else {
            if (K == 1) czr[b][0] += v[0] * cxa[0];
            if (K == 2) czr[b][1] += v[0] * cxa[1];
            if (K == 3) czr[b][2] += v[0] * cxa[2];
            if (K == 4) czr[b][0] += v[1] * cxa[3];
            if (K == 5) czr[b][1] += v[1] * cxb[0];
            if (K == 6) czr[b][2] += v[1] * cxb[1];
            if (K == 7) czr[b][0] += v[2] * cxb[2];
            if (K == 8) czr[b][1] += v[2] * cxb[3];
            if (K == 9) czr[b][2] += v[2] * cxc[0];
            if (K == 10) czr[b][0] += v[3] * cxc[1];
            if (K == 11) czr[b][1] += v[3] * cxc[2];
            if (K == 12) czr[b][2] += v[3] * cxc[3];
          }
        };
        // once per run: the run's sum over u coupled with the harmonics into the channel owner's message element
        auto finish = [&](auto bc) {
          constexpr int b = decltype(bc)::value;
          if (ABL & 1) return;
          const float cz0 = czr[b][0], cz1 = czr[b][1], cz2 = czr[b][2];
          if (TYPE == PT_SS) oacc[b][0] += cz0 * S[b][0];
          else if (TYPE == PT_SV) { oacc[b][0] += cz0 * S[b][0]; oacc[b][1] += cz0 * S[b][1]; oacc[b][2] += cz0 * S[b][2]; }
          else if (TYPE == PT_VS) { oacc[b][0] += cz0 * S[b][0]; oacc[b][1] += cz1 * S[b][0]; oacc[b][2] += cz2 * S[b][0]; }
          else if (TYPE == PT_VVS) { oacc[b][0] += cz0 * S[b][0]; oacc[b][0] += cz1 * S[b][1]; oacc[b][0] += cz2 * S[b][2]; }
          else if (TYPE == PT_VVV) {
            oacc[b][0] += cz1 * S[b][2]; oacc[b][1] += cz2 * S[b][0]; oacc[b][2] += cz0 * S[b][1];
            oacc[b][0] -= cz2 * S[b][1]; oacc[b][1] -= cz0 * S[b][2]; oacc[b][2] -= cz1 * S[b][0];
          } else {   // PT_VTV: rows (m0 m1 m2 | m1 m3 m4 | m2 m4 m5)
            oacc[b][0] += Mv[b][0] * cz0; oacc[b][1] += Mv[b][1] * cz0; oacc[b][2] += Mv[b][2] * cz0;
            oacc[b][0] += Mv[b][1] * cz1; oacc[b][1] += Mv[b][3] * cz1; oacc[b][2] += Mv[b][4] * cz1;
            oacc[b][0] += Mv[b][2] * cz2; oacc[b][1] += Mv[b][4] * cz2; oacc[b][2] += Mv[b][5] * cz2;
          }
        };
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        // what travels behind MFMA m of a tile.  Previous tile: block 0's contraction in slots 0 (read) and 3..14 (one operation
        // each), block 1's in 12 and 15..26.  This tile: the next tile's chain start (bias x edge factor) in 26 / 27, block 0's sum
        // in 27 (block 1's closes the tile)
        auto travel = [&](auto mc, const float* xp) {
          constexpr int m = decltype(mc)::value;
          if constexpr (m == 0) cop(B0{}, std::integral_constant<int, 0>{}, xp);
          if constexpr (m >= 3 && m <= 14) cop(B0{}, std::integral_constant<int, m - 2>{}, xp);
          if constexpr (m == 12) cop(B1{}, std::integral_constant<int, 0>{}, xp);
          if constexpr (m >= 15 && m <= 26) cop(B1{}, std::integral_constant<int, m - 14>{}, xp);
          if constexpr (m == 26) scale4(accN[0], bias_n, se[0]);
          if constexpr (m == 27) { scale4(accN[1], bias_n, se[1]); sumr4(accp[0], accS[0], accB[0]); }
        };
        // tile i carries the contraction of tile i - 1; the run's first tile carries one of zeros (same code, no second copy
        // of the loop body for the register allocator to fit)
        const float* xp = xs_lane + xo;
        accp[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; accp[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nt; ++i) {
          tile(tile0 + i, [&](auto mc) { travel(mc, xp); });
          if (i) xp += x_step;
        }
        // the run's last tile is contracted in the open (its successor belongs to another path type)
        __builtin_amdgcn_sched_barrier(0);
#define COPS(b) cop(b, std::integral_constant<int, 0>{}, xp); cop(b, std::integral_constant<int, 1>{}, xp); cop(b, std::integral_constant<int, 2>{}, xp); \
        cop(b, std::integral_constant<int, 3>{}, xp); cop(b, std::integral_constant<int, 4>{}, xp); cop(b, std::integral_constant<int, 5>{}, xp); \
        cop(b, std::integral_constant<int, 6>{}, xp); cop(b, std::integral_constant<int, 7>{}, xp); cop(b, std::integral_constant<int, 8>{}, xp); \
        cop(b, std::integral_constant<int, 9>{}, xp); cop(b, std::integral_constant<int, 10>{}, xp); cop(b, std::integral_constant<int, 11>{}, xp); \
        cop(b, std::integral_constant<int, 12>{}, xp); finish(b);
        COPS(B0{}) COPS(B1{})
#undef COPS
        if (flags & 2) {   // last run of the channel group: this lane owns msg[e][oo .. oo + (VOUT ? 3 : 1)); the edge's factor comes off
          if (oo < D_out) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
              if (e0 + 16 * b + n < E) {
                float* op = d.msg + (size_t)(e0 + 16 * b + n) * D_out + oo;
                op[0] = oacc[b][0] * ue[b];
                if (VOUT) { op[1] = oacc[b][1] * ue[b]; op[2] = oacc[b][2] * ue[b]; }
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
#undef M32
#undef MT32
#undef MT16
#undef HB
#undef TB
  }
  if (a.trace && blockIdx.x == 0 && tid == 0) { a.trace[2] = __builtin_readcyclecounter(); a.trace[3] = __builtin_amdgcn_s_memrealtime(); }
}

void launch_conv2h(const Conv2Args& a, hipStream_t st) {
  static int no_split = getenv("DBFR_CONV2_NOSPLIT") ? atoi(getenv("DBFR_CONV2_NOSPLIT")) : 0;
  constexpr int NW = 8;
  const size_t lds = CH_RING_BYTES + (size_t)NW * C2_WAVE_FLOATS * sizeof(float);
  const int n_cu = dbfr_current_cu_count();
  Conv2Args b = a;
  b.skew = 0;
  b.run_barrier = 0;
  b.no_split = no_split;
  // (the LDS attribute is set on every launch: it is per device, and a process may drive several)
#define V(x) { if (dbfr_launch_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2h<NW, x>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "k_conv2h: hipFuncSetAttribute(MaxDynamicSharedMemorySize)")) return; \
               hipLaunchKernelGGL((k_conv2h<NW, x>), dim3(n_cu), dim3(64 * NW), lds, st, b); return; }
#ifdef DBFR_DEV_VARIANTS   // developer build (DBFR_BUILD_DEV=1 python -m diffbindfr_amd.build): the ABL variants of the kernel's header comment
  static int abl = getenv("DBFR_CONV2H_ABL") ? atoi(getenv("DBFR_CONV2H_ABL")) : 0;
  if (abl == 1) V(1) if (abl == 2) V(2) if (abl == 3) V(3) if (abl == 4) V(4) if (abl == 7) V(7) if (abl == 8) V(8) if (abl == 16) V(16)
  if (abl == 32) V(32) if (abl == 64) V(64) if (abl == 256) V(256) if (abl == 512) V(512)
#endif
  for (int i = 0; i < a.n_conv; ++i)
    if (a.c[i].w.W2rinv || a.c[i].w.W1rinv) V(1024)            // a conv of this launch carries per-row factors
  V(0)
#undef V
}
