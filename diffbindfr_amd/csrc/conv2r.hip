// Fused tensor-product convolution, split-bf16 GEMM, W2 through an LDS ring (round 2; since round 3 the range-guard fall-back of k_conv2h).
//
// Arithmetic: every fp32 operand of the radial MLP's 144 x W GEMM is cut into three bf16 pieces, a = a1 + a2 + a3 (round-to-nearest each
// time; 3 x 8 significand bits hold all 24 bits of an fp32 number, so the sum is exact and bf16 has fp32's exponent range: no scaling).  W2 is
// cut once at model creation (api.cpp pack_conv2), the hidden layer h in registers right after its ReLU.  A product a b = sum_ij a_i b_j; each
// a_i b_j (8 x 8 bits) is exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16; the six partial products with i + j <= 4 are evaluated,
// smallest first, the three dropped ones are below 2^-23 |a b| (tools/exp/split_bf16.hip: error vs fp64 one third of the fp32 instruction's).
// k = 144 = 4 k-steps of 32 + 16 on the x16 instruction; the hidden layer's MFMA result registers are, piece by piece, the B operand of the W2
// tiles (step s, lane group g, slot j = hidden unit 16 (2 s + (j >> 2)) + 4 g + (j & 3)): no transpose, no LDS.  The first GEMM (144 x 144)
// stays on the fp32 instruction.  (Round 2 also shipped k_conv2s, the same arithmetic with every wave fetching its own copy of the W2 pieces
// through the vector L1 -- ~10 % slower, the L1 80 % busy with eight identical 13.5-KiB fetches per tile; retired in round 5, git history has it.)
//
// ONE copy of a tile enters the CU:
//   * an LDS ring of five slots, one per k-step of a tile (4 x three 1-KiB pieces, the last 16 k, + the tile's 16 bias
//     values), laid out like the tile in memory.  It is filled by the waves themselves in 512-byte shares: one 8-byte load per
//     lane into a staging register pair, one ds_write_b64 two k-steps later (why not LDS-DMA, and why without a single
//     branch: at `fetch` / `put` below);
//   * a wave keeps only TWO k-steps of W2 fragments in registers (the one its MFMAs read, the next one arriving from the ring
//     by ds_read_b128, lane-linear = conflict-free) instead of a whole tile, which pays for a second accumulator set:
//     both edge blocks run through a k-step together (two accumulator chains, fragments read once per tile), and the
//     contraction of tile i - 1 is hand-interleaved into the issue gaps of tile i's 60 MFMAs (two waves per SIMD share the matrix
//     pipe: a wave has to hide its own vector work);
//   * slot s is read by everybody during k-step s - 1, re-filled with the next tile's k-step s during k-step s + 2 and read
//     again two k-steps later; an s_barrier at the start of k-steps 0, 2 and 4 is all the ordering that needs (`turn` below).
//     Three barriers per tile, rarely waited for: the waves of a workgroup do identical work.
// Everything else (unit queue, hidden layer in registers, x rows / harmonics in wave-private LDS, channel-owner accumulation,
// tail split, bitwise independence from the unit -> workgroup assignment) is k_conv2's (conv2.hip).
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
#define C3_RING_BYTES 14336         // one tile (13824 B) + its 16 bias values, padded
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

template <int NW, int ABL = 0>   // ABL (developer, wrong results): 1 no contraction, 2 no ring barriers, 4 no ring filling, 32 no LDS wait in front of the barriers, 64 unit prologue only, 128 timeline stamps (DBFR_CONV2_TRACE), 8 no hazard nop
__global__ __launch_bounds__(64 * NW, 2) void k_conv2r(Conv2Args a) {
  constexpr int K = 144, KT = 9;
  constexpr int EPB = 32 * NW;                       // edges per block (unit)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_unit[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const char* ring = reinterpret_cast<const char*>(lds);          // [C3_RING_BYTES] first: inside the 16-bit offset of ds_read
  const unsigned ring_lds = (unsigned)(size_t)ring;               // its LDS byte address (low half of the flat address)
  float* xs = lds + C3_RING_BYTES / 4 + wave * C2_WAVE_FLOATS;    // [32][C2_XLD]
  float* shs = xs + 32 * C2_XLD;                     // [32][10]
  float* ms = shs + 32 * 10;                         // [32][8]
  int* s_gth = reinterpret_cast<int*>(ms + 32 * 8);  // [32]
  // developer timeline (ABL 128): stamps (tag << 56 | shader clock) of workgroup 0, one list per wave (tools/exp/ring_trace.py)
  int tr_n = 0;
  auto stamp = [&](unsigned long long tag) {
    if ((ABL & 128) && a.trace && blockIdx.x == 0 && lane == 0 && tr_n < C2_TRACE_CAP)
      a.trace[(size_t)wave * C2_TRACE_CAP + tr_n++] = (tag << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull);
  };

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
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // nothing of the last unit still reads or arms the ring
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
    // Filling the ring: every k-step slot (three 1-KiB pieces) is cut into six 512-byte shares; wave w moves share w % 6 of
    // every slot (piece >> 1, half & 1: one 8-byte load per lane into a staging register pair, one ds_write_b64 three k-steps
    // later), share w % 3 of the last-16-k slot, and the tile's 16 bias values -- waves 6, 7 (and 3..7 on the short slot)
    // duplicate a neighbour's share: same bytes to the same place.  No branch around any of it, on purpose: behind a
    // wave-uniform `if` hipcc's wait-count pass loses track and parks an s_waitcnt vmcnt(0) in front of every ds_write, i.e.
    // the wave sits out the load it has just issued.  Plain loads rather than LDS-DMA for the same kind of reason: a
    // global_load_lds piece costs its issuing wave 60-185 cycles next to MFMAs (14 a tile: measured 40 % of the sweep).
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.W2s, 0, d.w.n_tiles * C3_TILE_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.b2q, 0, d.w.n_tiles * 64, 0x00020000);
    const int sh6 = wv % 6, sh_i = sh6 >> 1, v8 = lane * 8 + (sh6 & 1) * 512, sh3 = wv % 3;
    const int vb4 = lane < 16 ? lane * 4 : 0x40000000;   // lanes 16..63: out of the buffer's range, the load returns 0 ...
    const int wb4 = lane < 16 ? lane * 4 : 64 + lane * 4; // ... and lands in the padding behind the bias values
    char* ringw = const_cast<char*>(ring);
    u32x2 stg[5];
    unsigned stgb = 0;
    auto fetch = [&](auto sc, int tile) {
      constexpr int s = decltype(sc)::value;
      if (ABL & 4) return;
      if (s < 4) stg[s] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rW, v8, tile * C3_TILE_BYTES + (sh_i * 4 + s) * 1024, 0));
      else stg[4] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rW, lane * 8, tile * C3_TILE_BYTES + C3_TAIL_OFF + sh3 * 512, 0));
      if (s == 0) stgb = __builtin_amdgcn_raw_buffer_load_b32(rB, vb4, tile * 64, 0);
    };
    auto put = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (ABL & 4) return;
      if (s < 4) *reinterpret_cast<u32x2*>(ringw + (sh_i * 4 + s) * 1024 + v8) = stg[s];
      else *reinterpret_cast<u32x2*>(ringw + C3_TAIL_OFF + sh3 * 512 + lane * 8) = stg[4];
      if (s == 0) *reinterpret_cast<unsigned*>(ringw + C3_TILE_BYTES + wb4) = stgb;
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    if (r_begin < r_end) {   // the part's first tile travels while the hidden layer is computed
      fetch(I0{}, t_first); fetch(I1{}, t_first); fetch(I2{}, t_first); fetch(I3{}, t_first); fetch(I4{}, t_first);
    }

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
    u32x4 Bh[2][3][4];                                 // h pieces: [edge block][piece][k-step of 32] = 8 bf16 each, 96 VGPRs
    u32x2 Bt[2][3];                                    // ... and of the last 16 k (v_mfma_f32_16x16x16_bf16): 4 bf16 each, 12 VGPRs
    {
      const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w.W1p, 0, KT * KT * 1024, 0x00020000);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
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
    {   // lane L moves the 16-byte chunks (L >> 5), (L >> 5) + 2, ... of row L & 31: all loads out before the first store, no
        // index arithmetic (as `for (i = lane; i < 32 * d4; i += 64)` hipcc made every chunk a division, an LDS look-up, two
        // scalar loads, a global load and its store in strict sequence: ~10 us per unit).  Chunks past the row's end repeat its
        // last one: same bytes to the same place.
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

    stamp(0x01);                                         // prologue done
    if (ABL & 64) {   // developer: the unit prologue alone
      asm volatile("" ::"v"(Bh[0][0][0]), "v"(Bh[1][2][3]), "v"(Bt[1][2]));
      continue;
    }
    // ---- the W2 row tiles of this part, run by run (channel-owner order, api.cpp pack_conv2)
    float oacc[2][3];
#pragma unroll
    for (int b = 0; b < 2; ++b) oacc[b][0] = oacc[b][1] = oacc[b][2] = 0.f;
    const float* xs_lane = xs + n * C2_XLD;          // + 16 b C2_XLD per edge block
    const float* sh_lane = shs + n * 10;
    bf16x8 FA[3], FB[3];                               // W2 pieces of two k-steps: 24 VGPRs
    s16x4 FT[3];                                       // ... and of the last 16 k: 6
    f32x4 bias_r;                                      // bias of the tile whose first MFMA comes next
    f32x4 acc[2], accp[2];                             // this tile's accumulators; the previous tile's, being contracted
    auto rd_step = [&](auto jc, bf16x8 (&F)[3]) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int i = 0; i < 3; ++i) F[i] = *reinterpret_cast<const bf16x8*>(ring + (i * 4 + j) * 1024 + vW);
    };
    // k-step s ("turn"): my share of the next tile's k-step s sets out for its staging registers (slot s was read during k-step
    // s - 1), the share fetched two turns ago goes into its slot, to be read two turns from now, and the next k-step's fragments
    // are requested -- each of the three behind one of the turn's first MFMAs, so that a wave leaves a barrier straight into matrix
    // work (the waves of a workgroup pass the barriers together: whatever follows one is a bubble on every SIMD at once).
    // Turns 0, 2 and 4 open with "my LDS reads and writes are done" + s_barrier; that is enough: between the last read of a slot
    // (turn x) and its rewrite (turn x + 2), and between the rewrite and the next read (turn x + 4), there is always one of the three.
    auto turn = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (s % 2 == 0) {
        __builtin_amdgcn_sched_barrier(0);
        stamp(0x10 + s);                                 // arrival at the turn
        if (!(ABL & 32)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(0x20 + s);                                 // my LDS operations are done
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
        stamp(0x30 + s);                                 // everybody is here
        __builtin_amdgcn_sched_barrier(0);
      } else if (ABL & 128) {
        __builtin_amdgcn_sched_barrier(0);
        stamp(0x10 + s);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#define X32(b, ai, hi, F, s) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[ai], __builtin_bit_cast(bf16x8, Bh[b][hi][s]), acc[b], 0, 0, 0)
#define X16(b, ai, hi) acc[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(FT[ai], __builtin_bit_cast(s16x4, Bt[b][hi]), acc[b], 0, 0, 0)
#define SLOT(m) do { op(std::integral_constant<int, (m)>{}); __builtin_amdgcn_sched_barrier(0); } while (0)
    // one tile: 60 MFMAs = 5 k-steps x 6 partial products (smallest first) x 2 edge blocks, `op(m)` = what travels behind MFMA m
    auto tile = [&](int t_next, auto&& op) {
      // e0 / e1 / e2: the turn's fetch, put and fragment request, behind MFMA 0 / 1 / 2 of the k-step
      auto kstep = [&](auto sc, bf16x8 (&F)[3], auto&& e0, auto&& e1, auto&& e2) {
        constexpr int s = decltype(sc)::value;
        X32(0, 2, 0, F, s); e0(); SLOT(12 * s + 0); X32(1, 2, 0, F, s); e1(); SLOT(12 * s + 1);
        X32(0, 1, 1, F, s); e2(); SLOT(12 * s + 2); X32(1, 1, 1, F, s); SLOT(12 * s + 3);
        X32(0, 0, 2, F, s); SLOT(12 * s + 4); X32(1, 0, 2, F, s); SLOT(12 * s + 5);
        X32(0, 1, 0, F, s); SLOT(12 * s + 6); X32(1, 1, 0, F, s); SLOT(12 * s + 7);
        X32(0, 0, 1, F, s); SLOT(12 * s + 8); X32(1, 0, 1, F, s); SLOT(12 * s + 9);
        X32(0, 0, 0, F, s); SLOT(12 * s + 10); X32(1, 0, 0, F, s); SLOT(12 * s + 11);
      };
      turn(I0{});
      acc[0] = bias_r; acc[1] = bias_r;
      kstep(I0{}, FA, [&] { fetch(I0{}, t_next); }, [&] { put(I3{}); }, [&] { rd_step(I1{}, FB); });
      turn(I1{});
      kstep(I1{}, FB, [&] { fetch(I1{}, t_next); }, [&] { put(I4{}); }, [&] { rd_step(I2{}, FA); });
      turn(I2{});
      kstep(I2{}, FA, [&] { fetch(I2{}, t_next); }, [&] { put(I0{}); }, [&] { rd_step(I3{}, FB); });
      turn(I3{});
      kstep(I3{}, FB, [&] { fetch(I3{}, t_next); }, [&] { put(I1{}); }, [&] {
#pragma unroll
        for (int i = 0; i < 3; ++i) FT[i] = *reinterpret_cast<const s16x4*>(ring + C3_TAIL_OFF + i * 512 + lane * 8);
      });
      turn(I4{});
      // k = 128..143 on v_mfma_f32_16x16x16_bf16.  Hazard (MI355X + ROCm 7.2, first met in round 2): an x16 MFMA taking as SrcC an
      // accumulator an x32 MFMA has just written reads stale data; 16 wait states put any pass count behind us.
      asm volatile("s_nop 15");
      __builtin_amdgcn_sched_barrier(0);
      X16(0, 2, 0); fetch(I4{}, t_next); SLOT(48); X16(1, 2, 0); put(I2{}); SLOT(49);
      X16(0, 1, 1); rd_step(I0{}, FA);                   // the NEXT tile's first k-step and bias
      bias_r = *reinterpret_cast<const f32x4*>(ring + C3_TILE_BYTES + vB);
      SLOT(50); X16(1, 1, 1); SLOT(51);
      X16(0, 0, 2); SLOT(52); X16(1, 0, 2); SLOT(53);
      X16(0, 1, 0); SLOT(54); X16(1, 1, 0); SLOT(55);
      X16(0, 0, 1); SLOT(56); X16(1, 0, 1); SLOT(57);
      X16(0, 0, 0); SLOT(58); X16(1, 0, 0); SLOT(59);
      accp[0] = acc[0]; accp[1] = acc[1];
      __builtin_amdgcn_sched_barrier(0);
      stamp(0x40);                                       // tile done
    };
#undef SLOT
    int x_phase = 0;
    if (r_begin < r_end) {                             // the part's first tile into the ring (its shares left before the hidden layer)
      put(I0{}); put(I1{}); put(I2{}); put(I3{}); put(I4{});   // (stg[3], stg[4] keep these shares: the first two turns write them once more)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      rd_step(I0{}, FA);
      bias_r = *reinterpret_cast<const f32x4*>(ring + C3_TILE_BYTES + vB);
    }
    for (int r = r_begin; r < r_end; ++r) {
      const RunDesc rd = d.w.runs[r];
      const int tile0 = rd.tile0_n & 0xfffff, nt = rd.tile0_n >> 20;
      const int type = rd.meta & 15, flags = (rd.meta >> 4) & 3, sh_off = (rd.meta >> 8) & 15, x_step = (rd.meta >> 12) & 0xff;
      const int xph = (rd.meta >> 20) & 1;
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
        float S[2][NSV ? NSV : 1];
        if (TYPE != PT_VTV) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const float* sp = sh_lane + 160 * b + sh_off;
#pragma unroll
            for (int k = 0; k < NSV; ++k) S[b][k] = sp[k];
          }
        }
        // the contraction of one edge block of the PREVIOUS tile (accp) as micro-operations: K = 0 the first LDS reads, K >= 1
        // one FMA each (K beyond the type's count: nothing).  State lives in cx* / cz* between the operations.
        f32x4 cxa, cxb, cxc, cma;
        float2 cmb;
        float cz0, cz1, cz2;
        auto cop = [&](auto bc, auto kc, const float* xp) {
          constexpr int b = decltype(bc)::value;
          constexpr int K = decltype(kc)::value;
          if (ABL & 1) { if (K == 1) asm volatile("" ::"v"(accp[b])); return; }
          const f32x4 v = accp[b];
          const f32x4* x4 = reinterpret_cast<const f32x4*>(xp + 16 * b * C2_XLD);   // [u0..u0+3][3] = 12 consecutive floats (VIN)
          if (VIN && K == 2) cxb = x4[1];            // (each quad requested a few operations ahead of its first use)
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
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        // what travels behind MFMA m of a tile: block 0 of the previous tile in slots 0 (reads) and 5..26, block 1 in 28 and 33..54
        auto travel = [&](auto mc, const float* xp) {
          constexpr int m = decltype(mc)::value;
          if constexpr (m == 0) cop(B0{}, std::integral_constant<int, 0>{}, xp);
          else if constexpr (m >= 5 && m <= 26) cop(B0{}, std::integral_constant<int, m - 4>{}, xp);
          else if constexpr (m == 28) cop(B1{}, std::integral_constant<int, 0>{}, xp);
          else if constexpr (m >= 33 && m <= 54) cop(B1{}, std::integral_constant<int, m - 32>{}, xp);
        };
        // tile i carries the contraction of tile i - 1; the run's first tile carries one of zeros (same code, no second copy
        // of the loop body for the register allocator to fit)
        const float* xp = xs_lane + xo;
        accp[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; accp[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nt; ++i) {
          const int t = tile0 + i;
          tile(t < t_last ? t + 1 : t, [&](auto mc) { travel(mc, xp); });
          if (i) xp += x_step;
        }
        // the run's last tile is contracted in the open (its successor belongs to another path type).  Its accumulators were
        // last written by x16 MFMAs a few instructions ago: hipcc's wait states for that opcode are not to be relied on (see
        // the hazard above), and nothing but LDS latency stands between them and the first FMA that reads them
        __builtin_amdgcn_sched_barrier(0);
        stamp(0x50);                                     // run's last contraction starts
        asm volatile("s_nop 15");
        __builtin_amdgcn_sched_barrier(0);
#define COPS(b) cop(b, std::integral_constant<int, 0>{}, xp); cop(b, std::integral_constant<int, 1>{}, xp); cop(b, std::integral_constant<int, 2>{}, xp); \
        cop(b, std::integral_constant<int, 3>{}, xp); cop(b, std::integral_constant<int, 4>{}, xp); cop(b, std::integral_constant<int, 5>{}, xp); \
        cop(b, std::integral_constant<int, 6>{}, xp); cop(b, std::integral_constant<int, 7>{}, xp); cop(b, std::integral_constant<int, 8>{}, xp); \
        cop(b, std::integral_constant<int, 9>{}, xp); cop(b, std::integral_constant<int, 10>{}, xp); cop(b, std::integral_constant<int, 11>{}, xp); \
        cop(b, std::integral_constant<int, 12>{}, xp); cop(b, std::integral_constant<int, 13>{}, xp); cop(b, std::integral_constant<int, 14>{}, xp); \
        cop(b, std::integral_constant<int, 15>{}, xp); cop(b, std::integral_constant<int, 16>{}, xp); cop(b, std::integral_constant<int, 17>{}, xp); \
        cop(b, std::integral_constant<int, 18>{}, xp); cop(b, std::integral_constant<int, 19>{}, xp); cop(b, std::integral_constant<int, 20>{}, xp); \
        cop(b, std::integral_constant<int, 21>{}, xp);
        COPS(B0{}) COPS(B1{})
#undef COPS
        stamp(0x51);
        if (flags & 2) {   // last run of the channel group: this lane owns msg[e][oo .. oo + (VOUT ? 3 : 1))
          if (oo < D_out) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
              if (e0 + 16 * b + n < E) {
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
#undef X32
#undef X16
  }
  // ---- the last workgroup to leave re-arms the queue for the next launch
  if (tid == 0) {
    const int dn = atomicAdd(a.queue + 1, 1);
    if (dn == n_wg - 1) { a.queue[0] = 0; a.queue[1] = 0; __threadfence(); }
  }
}

void launch_conv2r(const Conv2Args& a, hipStream_t st) {
  static int no_split = getenv("DBFR_CONV2_NOSPLIT") ? atoi(getenv("DBFR_CONV2_NOSPLIT")) : 0;
  constexpr int NW = 8;
  const size_t lds = C3_RING_BYTES + (size_t)NW * C2_WAVE_FLOATS * sizeof(float);
  const int n_cu = dbfr_current_cu_count();
  Conv2Args b = a;
  b.skew = 0;
  b.run_barrier = 0;
  b.no_split = no_split;
  // (the LDS attribute is set on every launch: it is per device, and a process may drive several)
#define V(x) { if (dbfr_launch_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv2r<NW, x>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "k_conv2r: hipFuncSetAttribute(MaxDynamicSharedMemorySize)")) return; \
               hipLaunchKernelGGL((k_conv2r<NW, x>), dim3(n_cu), dim3(64 * NW), lds, st, b); return; }
#ifdef DBFR_DEV_VARIANTS   // developer build (DBFR_BUILD_DEV=1): timing-only ablations (wrong results)
  static int abl = getenv("DBFR_CONV2R_ABL") ? atoi(getenv("DBFR_CONV2R_ABL")) : 0;
  if (abl == 1) V(1) if (abl == 2) V(2) if (abl == 34) V(34) if (abl == 64) V(64) if (abl == 128) V(128)
#endif
  V(0)
#undef V
}
