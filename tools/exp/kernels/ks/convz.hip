// Reduce-first form of the SCALAR-OUTPUT tensor-product paths of a K = 144 conv (DBFR_GEMM_REDUCE_FIRST) -- round 6, "k-split" form.
// Algorithm, scaling and message interface as in round 5's kernel (diffbindfr_amd/csrc/convz.hip of commit e46b983, docs/kernels/conv_reduce_first.md):
//     Z[t, c, k] = sum_{e -> t} y[e, c] h'[e, k]     (step A, per chunk of <= 32 edges / <= 4 targets, segment = target inside a chunk)
//     out_sum[t, o] = sum_{c,k} Z[t, c, k] W2'[c, k, o]   (step B, columns = segments)
// What changed is how the work is laid over a compute unit.  Measured in round 6 (profiles/TUNING_r6.md): the tile loop is bound by the stream of W2'
// fragments out of L2 -- 48 KB per (c, k) tile and workgroup at ~26 B per cycle, whatever the waves do meanwhile (a second tile in flight doubles the
// latency, not the rate) -- so the only lever is MORE COLUMNS PER FRAGMENT TILE, i.e. more chunks resident per workgroup, and a chunk's state is its
// hidden layer H (80 registers per lane).  Here:
//   * SIXTEEN waves in two roles: twelve CHUNK waves (one chunk each: hidden layer, Y, step A, the cutting of Z into fp16 pieces) and four COLUMN waves
//     (two k-steps of step B each: W2' fragments, accumulators of 48 columns) -- 128 registers per wave, four waves per SIMD;
//   * the k range in TWO PASSES (k tiles 0..4, then 5..8 and the bias tile): a chunk wave holds half of H at a time (40 registers), computes the other half
//     at the start of the second pass (W1h half by half into LDS, the inputs loaded again: they are L2 hits) and the partial sums of the second pass are
//     ADDED to those the first one stored (same workgroup, fixed order: no atomics, batch-independent);
//   * 12 chunks = 48 columns per W2' tile instead of 32: 1.5 x fewer fragment bytes per edge.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CH_TILE_BYTES 9280          // W1h tile format of conv2h.hip
#define CH_TAIL_OFF 8192
#define CH_BIAS_OFF 9216

#define CZ_NWA 12                   // chunk waves = chunks per unit
#define CZ_NWB 4                    // column waves
#define CZ_NW (CZ_NWA + CZ_NWB)
#define CZ_KS (8 / CZ_NWB)          // k-steps of 32 (of the 256 (c, k) values of a tile) per column wave
#define CZ_KTP 5                    // k tiles per pass

// Z in LDS, already cut into fp16 pieces by the chunk wave that computed it.  Per buffer and column block two planes (hi, lo), each
// [k-step v: 1040 B][lane group g'': 256 B][column: 16 B = (k_local v: c 4g''..4g''+3 | k_local v + 8: the same c)].  Step A's lane (k_local n, group g) writes
// its four values c_local = 4 g + q as one 8-byte store per plane to (v = n & 7, g'' = g, half n >> 3); step B's lane (column, g'') reads the 16 bytes of its
// k-step from each plane: they ARE the B operand.  The 1040-byte stride (4 dwords mod 32 banks) makes the stores conflict-free, the reads are contiguous.
// Two buffers 64 KiB apart (a tile's buffer = the parity of a running tile count: five tiles per c tile; the toggle is one XOR).
#define CZ_VSTRIDE 1040
#define CZ_PLANE (8 * CZ_VSTRIDE)
#define CZ_CB (2 * CZ_PLANE)        // one column block: hi plane, lo plane
#define CZ_NCB 3                    // column blocks of 16: the <= 12 x CZ_MAXSEG segments of the workgroup's chunks take consecutive columns
#define CZ_BUF (CZ_NCB * CZ_CB)     // one (c, k) tile: all column blocks (49 920 B)
#define CZ_TOG 65536                // byte distance of the two buffers
#define CZ_ZBYTES (CZ_TOG + CZ_BUF)
#define CZ_WAVE_FLOATS (32 * 12 + 32 + 32 + 32 + 32 + 32 + CZ_MAXSEG * 16)   // per chunk wave: harmonics [32][12] | sa | ua | gather row offsets | segment ids | first slot of segment j | masks [CZ_MAXSEG segments][4 lane groups][4 dwords]
#define CZ_BLOCK_INTS (CZ_NWA + 2 * 16 * CZ_NCB + CZ_MAXCT * 16)                 // b_nseg | b_col_edge | b_col_inv | b_cdesc
#define CZ_ZSCALE (-20)             // |Z| <= 32 edges x 2^15 x 2^15 = 2^35 -> 2^15

__device__ __forceinline__ void cz_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector((f32x2){x0, x1}, f16x2);
  const f16x2 l = __builtin_convertvector((f32x2){x0 - (float)h[0], x1 - (float)h[1]}, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ float cz_wave_max(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <int I, int N, typename F>
__device__ __forceinline__ void cz_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); cz_static_for<I + 1, N>(f); }
}

#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define SL __builtin_amdgcn_sched_barrier(0)

// Both roles walk the same sequence of workgroup barriers: per unit and pass P1 (the pass's half of W1h is in LDS; the first pass: the c-tile descriptors
// too), P2 (hidden layer done: the Z buffers are free; the first pass: segment counts written), then per output irrep B3 (first Z tile written), one B4 per
// (c, k) tile, B5 (partial sums written), B6 (stored).
template <int ABL = 0>   // ABL (developer, wrong results): 4 no step A, 8 no step B; 128 (right results): s_memtime timeline of one unit of workgroup 0
__global__ __launch_bounds__(64 * CZ_NW) void k_convz(ConvZArgs a) {
  constexpr int KT = 9, NWA = CZ_NWA, NWB = CZ_NWB, KS = CZ_KS, NCB = CZ_NCB, KTP = CZ_KTP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave: a scalar register)
  const int n = lane & 15, g = lane >> 4;
  char* zb = reinterpret_cast<char*>(lds);
  int* b_nseg = reinterpret_cast<int*>(lds + CZ_ZBYTES / 4 + NWA * CZ_WAVE_FLOATS);   // [NWA]
  int* b_col_edge = b_nseg + NWA;                           // [16 NCB] message row of the column's segment, -1: column unused
  float* b_col_inv = reinterpret_cast<float*>(b_col_edge + 16 * NCB);   // [16 NCB] takes the chunk's factors off (rewritten per pass)
  unsigned* b_cdesc = reinterpret_cast<unsigned*>(b_col_inv + 16 * NCB);   // [CZ_MAXCT][16] the conv's c-tile descriptors

  // developer timeline (ABL & 128, DBFR_CONVZ_DEBUG=<file>): s_memtime stamps, [wave][stamp] unsigned long long in a.dbg
  int tr_n = 0;
  bool tr_on = false;                                       // (workgroup 0's THIRD unit -- warm caches -- of the launches whose first conv has a.dbg_sel c tiles; 0: any)
  auto stamp = [&](int tag) {
    if constexpr ((ABL & 128) != 0) {
      if (a.dbg && tr_on && lane == 0 && tr_n < 500) {
        unsigned long long* t = reinterpret_cast<unsigned long long*>(a.dbg) + wave * 512;
        t[tr_n++] = (__builtin_amdgcn_s_memtime() << 8) | (unsigned)tag;
      }
    }
  };
  // ---- unit list: NWA chunks per unit, conv after conv
  int nch[4] = {0, 0, 0, 0}, nu[4] = {0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < a.n_conv) {
      const ConvZDesc& d = a.c[c];
      nch[c] = min(*d.n_chunks, d.max_chunks);
      nu[c] = (nch[c] + NWA - 1) / NWA;
    }
  const int N = nu[0] + nu[1] + nu[2] + nu[3];
  // an output irrep of a pass is complete: add the column waves' partial sums (B5 behind them), take the factors off, store into the segments' first rows (the
  // second pass: add to what the first one stored) -- all sixteen waves
  auto store_irrep = [&](const ConvZDesc& d, const ConvZ& W, int io, int ncb, int pass) {
    for (int idx = tid; idx < 16 * ncb * 48; idx += 64 * CZ_NW) {
      const int col = idx / 48, w = idx - 48 * col;
      const int e = b_col_edge[col];
      if (e >= 0) {
        const int r = w & 15;
        const float* rp = reinterpret_cast<const float*>(zb) + (((w >> 4) * NCB + (col >> 4)) * 4 + (r & 3)) * 64 + 16 * (r >> 2) + (col & 15);
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NWB; ++v) s += rp[v * (12 * NCB * 64)];
        float* mp = d.msg + (size_t)e * d.D_out + W.out_off[io] + w;
        const float v = s * b_col_inv[col] * W.rowinv[io * 48 + w];
        *mp = pass ? *mp + v : v;
      }
    }
  };

  if (wave < NWA) {
    // =============================================================== chunk waves: wave w owns chunk w of the unit
    float* wl = lds + CZ_ZBYTES / 4 + wave * CZ_WAVE_FLOATS;
    float* w_sh = wl;                                         // [32][12]
    float* w_sa = wl + 32 * 12;                               // [32] the edge's factor on its radial-MLP inputs
    float* w_ua = w_sa + 32;                                  // [32] ... and its inverse
    int* w_row = reinterpret_cast<int*>(w_ua + 32);           // [32] gth[e] * ldx
    int* w_seg = w_row + 32;                                  // [32] segment of the slot, -1: no edge
    int* w_first = w_seg + 32;                                // [32] first slot of segment j
    unsigned* w_mask = reinterpret_cast<unsigned*>(w_first + 32);   // [CZ_MAXSEG][4][4] segment j's mask on the A-operand registers of lane group g
    for (int unit = blockIdx.x; unit < N; unit += gridDim.x) {
      int c = 0, ul = unit;
      if (ul >= nu[0]) { ul -= nu[0]; c = 1; if (ul >= nu[1]) { ul -= nu[1]; c = 2; if (ul >= nu[2]) { ul -= nu[2]; c = 3; } } }
      const ConvZDesc& d = a.c[c];
      const ConvZ& W = d.w;
      const int E = min(*d.n_edges, d.max_edges);
      // ---- my chunk: edges [es, es + len)
      const int ch = ul * NWA + wave;
      int es = 0, len = 0, gidx = 0;
      if (ch < nch[c]) {                                       // (k_chunk_fill: <= 32 edges and <= CZ_MAXSEG targets of one graph)
        es = d.chunk_es[ch];
        const int gl = d.chunk_gl[ch];
        gidx = gl >> 6;
        len = min(min(gl & 63, 32), max(E - es, 0));
      }
      tr_on = (ABL & 128) && blockIdx.x == 0 && unit == 2 * (int)gridDim.x && (a.dbg_sel == 0 || a.dbg_sel == a.c[0].w.ct0[a.c[0].w.n_io - 1] + a.c[0].w.nct[a.c[0].w.n_io - 1]);
      if (tr_on) tr_n = 0;
      stamp(1);
      // ---- slots: lanes 0..31 own slot L (clamped to the chunk's last edge beyond its length; a chunk without edges reads nothing)
      const int sl = lane & 31;
      const bool have = len > 0;
      const int e_sl = have ? es + min(sl, len - 1) : 0;
      // (my edge of the two edge tiles of the hidden layer, and its radial-MLP rows: the indices are requested here, with the slots' loads)
      const int e_et0 = have ? es + min(n, len - 1) : 0, e_et1 = have ? es + min(16 + n, len - 1) : 0;
      int tgt_l = 0, gth_l = 0, i1_0 = 0, i1_1 = 0, i2_0 = 0, i2_1 = 0;
      float shv[9];
      float xmx = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) shv[k] = 0.f;
      if (have) {
        tgt_l = d.tgt[e_sl]; gth_l = d.gth[e_sl];
        i1_0 = d.idx1[e_et0]; i1_1 = d.idx1[e_et1]; i2_0 = d.idx2[e_et0]; i2_1 = d.idx2[e_et1];
        if (lane < 32) {
          const float* sp = d.sh + (size_t)e_sl * SH_LD;
#pragma unroll
          for (int k = 0; k < 9; ++k) shv[k] = sp[k];
        }
        if (d.xmax) xmx = d.xmax[gidx];                       // per graph, computed once per layer (k_row_absmax)
        else {
          const float* xr = d.x + (size_t)gth_l * d.ldx;
          const int d4 = d.ldx >> 2;
          for (int j = lane >> 5; j < d4; j += 2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * j);
            xmx = fmaxf(fmaxf(xmx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
          }
        }
      }
      const int tgt_prev = __shfl_up(tgt_l, 1);
      const bool is_first = lane < 32 && sl < len && (sl == 0 || tgt_l != tgt_prev);
      const unsigned firsts = (unsigned)__ballot(is_first);
      const int nseg = min(__popc(firsts), CZ_MAXSEG);         // (the chunk table holds no chunk with more)
      const int seg_l = (sl < len) ? __popc(firsts & (0xffffffffu >> (31 - sl))) - 1 : -1;
      float smx = 0.f;
      if (lane < 32) {
        w_row[sl] = gth_l * d.ldx;
        w_seg[sl] = seg_l;
        if (is_first) w_first[seg_l] = sl;
#pragma unroll
        for (int k = 0; k < 9; ++k) { w_sh[sl * 12 + k] = shv[k]; smx = fmaxf(smx, fabsf(shv[k])); }
        w_sh[sl * 12 + 9] = w_sh[sl * 12 + 10] = w_sh[sl * 12 + 11] = 0.f;
      }
      if (lane == 0) b_nseg[wave] = nseg;
      // (the scalar columns of the message rows that are not the first of their segment are NOT written: the reductions read those columns of flagged rows
      // only -- EdgeSet::seg_first, conv.hip row_sum)
      xmx = cz_wave_max(xmx); smx = cz_wave_max(smx);
      int ey = 0;                                              // y is multiplied by 2^ey: |y| <= 3 |x| |sh| < 2^(ex + es + 2) -> below 2^15
      if (xmx > 0.f && smx > 0.f) ey = max(-100, min(100, 13 - __builtin_amdgcn_frexp_expf(xmx) - __builtin_amdgcn_frexp_expf(smx)));
      ey = __builtin_amdgcn_readfirstlane(ey);
      const float sY = __builtin_amdgcn_ldexpf(1.f, ey);
      const int nseg_u = __builtin_amdgcn_readfirstlane(nseg);
      __builtin_amdgcn_wave_barrier();
      // ---- masks of my segments on the registers of the A operand (halves of a dword = two consecutive slots), for every lane group
      {                                                        // (segments the chunk does not have: all-zero masks)
        const int j = lane >> 4, gg = (lane >> 2) & 3, r = lane & 3, t0 = 2 * r;
        const int s0 = t0 < 4 ? 4 * gg + t0 : 16 + 4 * gg + (t0 - 4);
        w_mask[lane] = (w_seg[s0] == j ? 0x0000ffffu : 0u) | (w_seg[s0 + 1] == j ? 0xffff0000u : 0u);
      }
      stamp(2);
      const float zs = __builtin_amdgcn_ldexpf(1.f, CZ_ZSCALE);
      int cbase = 0, ncb = 0;
      char* za[CZ_MAXSEG];
      for (int pass = 0; pass < 2; ++pass) {
        // ---- hidden layer of this pass's unit tiles m = 5 pass + jj (jj < 5; the second pass has four, its fifth k tile carries the bias), transposed:
        // D[edge, unit] = sum_f a[edge, f] W1[unit, f]; A = the edge's inputs (cut per edge), B = W1h tiles from LDS; the two edge tiles one after the other
        u32x4 Hh[KTP][2];                                      // H pieces [k tile of the pass][hi, lo]: lane (unit n, group g), eight edges {4g..4g+3, 16+4g..16+4g+3}
        int eh = 0;                                            // the factor on h is 2^(15 - ehc + k1) = phi, also what the constant 1 of the bias becomes
        {
          float Hf[KTP][2][4];
          float hmx = 0.f;
          // raw radial-MLP inputs of my edge of an edge tile: [emb | tab1 row | tab2 row], 16 floats of each per lane group.  One edge tile at a time: the
          // inputs of both, the results so far and the fragment ring exceed the 128 registers of a wave (the second tile's loads are exposed: L2 hits)
          auto load_inputs = [&](int et, f32x4 (&Ba)[KT]) {
            const float* r0 = d.emb + (size_t)(et ? e_et1 : e_et0) * NS;
            const float* r1 = d.tab1 + (size_t)(et ? i1_1 : i1_0) * d.ld1;
            const float* r2 = d.tab2 + (size_t)(et ? i2_1 : i2_0) * d.ld2;
#pragma unroll
            for (int s4 = 0; s4 < KT; ++s4) {
              const float* src = s4 < 3 ? r0 : s4 < 6 ? r1 : r2;
              Ba[s4] = have ? *reinterpret_cast<const f32x4*>(src + 16 * (s4 % 3) + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          };
          f32x4 Ba0[KT];
          load_inputs(0, Ba0);
          __syncthreads();   // P1: this pass's half of W1h is in LDS (tile jj at jj x 9280), and the c-tile descriptors (column waves)
          stamp(4);
          cz_static_for<0, 2>([&](auto et_c) {
            constexpr int et = decltype(et_c)::value;
            f32x4 Ba1[KT];
            if constexpr (et == 1) load_inputs(1, Ba1);
            f32x4 (&Ba)[KT] = et == 0 ? Ba0 : Ba1;
            float amx = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < KT; ++s4)
#pragma unroll
              for (int q = 0; q < 4; ++q) amx = fmaxf(amx, fabsf(Ba[s4][q]));
            amx = fmaxf(amx, __shfl_xor(amx, 16));
            amx = fmaxf(amx, __shfl_xor(amx, 32));
            const int ja = max(-14, __builtin_amdgcn_frexp_expf(amx));
            const float sa = __builtin_amdgcn_ldexpf(1.f, 15 - ja);
            if (g == 0) { w_sa[16 * et + n] = sa; w_ua[16 * et + n] = __builtin_amdgcn_ldexpf(1.f, ja - 15); }
            u32x4 Ah[2][4];                                    // input pieces [hi, lo][k-step of 32]
            u32x4 Atc;                                         // last 16 k: [lo | hi]
#pragma unroll
            for (int s4 = 0; s4 < KT; ++s4) {
              unsigned hi0, lo0, hi1, lo1;
              cz_split2(Ba[s4][0] * sa, Ba[s4][1] * sa, hi0, lo0);
              cz_split2(Ba[s4][2] * sa, Ba[s4][3] * sa, hi1, lo1);
              if (s4 < 8) {
                Ah[0][s4 >> 1][2 * (s4 & 1)] = hi0; Ah[0][s4 >> 1][2 * (s4 & 1) + 1] = hi1;
                Ah[1][s4 >> 1][2 * (s4 & 1)] = lo0; Ah[1][s4 >> 1][2 * (s4 & 1) + 1] = lo1;
              } else {
                Atc = (u32x4){lo0, lo1, hi0, hi1};
              }
            }
            __builtin_amdgcn_wave_barrier();
            // my rows of a result tile are the edges 16 et + 4 g + q: their factors
            const f32x4 sar = *reinterpret_cast<const f32x4*>(w_sa + 16 * et + 4 * g), uar = *reinterpret_cast<const f32x4*>(w_ua + 16 * et + 4 * g);
            // W1 fragments: step i = 5 jj + s (s < 4: the k-step of 32 of tile jj, hi and lo piece; s = 4: the tile's last 16 k [hi | lo] and, in the second
            // register, my unit's bias) through a ring of three register pairs, read TWO steps ahead of their MFMAs
            u32x4 F[3][2];
            auto ldF = [&](auto ic) {
              constexpr int i = decltype(ic)::value, jj = i / 5, sx = i % 5;
              if constexpr (sx < 4) {
                F[i % 3][0] = *reinterpret_cast<const u32x4*>(zb + jj * CH_TILE_BYTES + sx * 1024 + lane * 16);
                F[i % 3][1] = *reinterpret_cast<const u32x4*>(zb + jj * CH_TILE_BYTES + (4 + sx) * 1024 + lane * 16);
              } else {
                F[i % 3][0] = *reinterpret_cast<const u32x4*>(zb + jj * CH_TILE_BYTES + CH_TAIL_OFF + lane * 16);
                F[i % 3][1][0] = *reinterpret_cast<const unsigned*>(zb + jj * CH_TILE_BYTES + CH_BIAS_OFF + n * 4);
              }
            };
            ldF(std::integral_constant<int, 0>{});
            ldF(std::integral_constant<int, 1>{});
            // three accumulators (the two small products, the large one), added behind the tile: no matrix instruction waits for the one before it
            f32x4 aS, aT, aL;
            cz_static_for<0, 5 * KTP>([&](auto ic) {
              constexpr int i = decltype(ic)::value, jj = i / 5, sx = i % 5;
              SL;                                              // (pins the step order: left to itself hipcc hoists the LDS reads of many steps and spills)
              if constexpr (i + 2 < 5 * KTP) ldF(std::integral_constant<int, (i + 2 < 5 * KTP ? i + 2 : 0)>{});
              if constexpr (sx == 0) { aS = (f32x4){0.f, 0.f, 0.f, 0.f}; aT = aS; aL = aS; }
              if constexpr (sx < 4) {
                const f16x8 whi = __builtin_bit_cast(f16x8, F[i % 3][0]), wlo = __builtin_bit_cast(f16x8, F[i % 3][1]);
                aS = MF(__builtin_bit_cast(f16x8, Ah[1][sx]), whi, aS);
                aT = MF(__builtin_bit_cast(f16x8, Ah[0][sx]), wlo, aT);
                aL = MF(__builtin_bit_cast(f16x8, Ah[0][sx]), whi, aL);
              } else {
                // k = 128..143: W tail fragment = [hi (4) | lo (4)], input tail = [lo | hi]: one x32 MFMA carries both small products; the large product
                // on the x32 instruction too, upper half of the operands zero (an x16 MFMA that takes as SrcC an accumulator an x32 MFMA has just
                // written reads stale data on MI355X + ROCm 7.2 -- DESIGN.md section 4, invariant 3)
                const u32x4 wt = F[i % 3][0];
                const f16x8 wf = __builtin_bit_cast(f16x8, wt);
                const f16x8 wh = __builtin_bit_cast(f16x8, (u32x4){wt[0], wt[1], 0u, 0u});
                const float bias = __builtin_bit_cast(float, F[i % 3][1][0]);
                const int m = 5 * pass + jj;                   // (the unit tile: the second pass's fifth does not exist -- zeros in, nothing kept)
                const float w1r = (W.W1rinv && m < KT) ? W.W1rinv[16 * m + n] : 1.f;
                aS = MF(__builtin_bit_cast(f16x8, Atc), wf, aS);
                aL = MF(__builtin_bit_cast(f16x8, (u32x4){Atc[2], Atc[3], 0u, 0u}), wh, aL);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float v = fmaxf((aS[q] + aT[q]) + aL[q] + bias * sar[q], 0.f) * (uar[q] * w1r);   // = 2^k1 h (w1r: my unit's row factor of W1h off)
                  Hf[jj][et][q] = v;
                  if (jj < 4 || pass == 0) hmx = fmaxf(hmx, v);
                }
              }
            });
          });
          hmx = cz_wave_max(hmx);
          // 2^k1 h < 2^eh; the factor 2^(15 - ehc) on it makes phi = 2^(15 - ehc + k1) on h: ehc >= k1 keeps phi, the image of the bias's 1, inside fp16
          eh = max(hmx > 0.f ? __builtin_amdgcn_frexp_expf(hmx) : W.k1, W.k1);
          eh = __builtin_amdgcn_readfirstlane(min(eh, W.k1 + 100));
          const float sH = __builtin_amdgcn_ldexpf(1.f, 15 - eh);
#pragma unroll
          for (int jj = 0; jj < KTP; ++jj) {
            unsigned h0, l0, h1, l1, h2, l2, h3, l3;
            cz_split2(Hf[jj][0][0] * sH, Hf[jj][0][1] * sH, h0, l0);
            cz_split2(Hf[jj][0][2] * sH, Hf[jj][0][3] * sH, h1, l1);
            cz_split2(Hf[jj][1][0] * sH, Hf[jj][1][1] * sH, h2, l2);
            cz_split2(Hf[jj][1][2] * sH, Hf[jj][1][3] * sH, h3, l3);
            Hh[jj][0] = (u32x4){h0, h1, h2, h3};
            Hh[jj][1] = (u32x4){l0, l1, l2, l3};
          }
        }
        const int ephi = 15 - eh + W.k1;                       // log2 of the factor on h
        if (pass) {   // the k tile 9 (the second pass's fifth): the constant 1 (x phi) of the bias in column 0, for the slots that hold an edge
          const _Float16 ph = (_Float16)__builtin_amdgcn_ldexpf(1.f, ephi);
          const unsigned short pb = __builtin_bit_cast(unsigned short, ph);
          u32x4 hb = {0u, 0u, 0u, 0u};
          if (n == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const int slot = t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4);
              if (slot < len) hb[t >> 1] |= (unsigned)pb << (16 * (t & 1));
            }
          }
          Hh[KTP - 1][0] = hb;
          Hh[KTP - 1][1] = (u32x4){0u, 0u, 0u, 0u};
        }
        stamp(3);
        __syncthreads();   // P2: the Z buffers are free again; the segment counts of the chunks and the cleared column table (first pass)
        if (pass == 0) {
          // ---- columns: the segments of the unit's chunks side by side (<= 12 x CZ_MAXSEG = 48 = three column blocks of step B)
          int total = 0;
#pragma unroll
          for (int v = 0; v < NWA; ++v) { const int cnt = b_nseg[v]; if (v < wave) cbase += cnt; total += cnt; }
          cbase = __builtin_amdgcn_readfirstlane(cbase); total = __builtin_amdgcn_readfirstlane(total);   // (wave-uniform)
          ncb = (total + 15) >> 4;                             // column blocks of 16 in step B: 0 (no edges) .. 3
          if (lane < nseg_u) b_col_edge[cbase + lane] = es + w_first[lane];
          // LDS addresses of my segments' Z columns in buffer 0, hi plane (lane part + column).  A slot without a segment computes zeros and writes them into
          // the last column: unused whenever such a slot exists (the unit then has fewer than 48 segments), and never stored
          char* zw0 = zb + (n & 7) * CZ_VSTRIDE + g * 256 + (n >> 3) * 8;
#pragma unroll
          for (int j = 0; j < CZ_MAXSEG; ++j) {
            const int col = cbase + j;
            za[j] = j < nseg_u ? zw0 + (col >> 4) * CZ_CB + (col & 15) * 16 : zw0 + (NCB - 1) * CZ_CB + 15 * 16;
          }
        }
        if (lane < nseg_u) b_col_inv[cbase + lane] = __builtin_amdgcn_ldexpf(1.f, -CZ_ZSCALE - ey - ephi);
        // Y of a c tile, MASKED per segment: lane (c = n, group g), my eight slots; Ym[j] = the pieces with the slots outside segment j zeroed -- the
        // A operand of step A for the five k tiles of the c tile and pass.  The x values are gathered straight into registers, at the top of the tile in
        // which the column waves work through the last k tile of the c tile before (the chunk waves have nothing else to do in it).
        u32x4 Ym[CZ_MAXSEG][2];
#pragma unroll
        for (int j = 0; j < CZ_MAXSEG; ++j) { Ym[j][0] = (u32x4){0u, 0u, 0u, 0u}; Ym[j][1] = (u32x4){0u, 0u, 0u, 0u}; }
        auto make_Y = [&](int ctg) {                           // ctg: c tile of the conv (over both irreps)
          if (nseg_u == 0) return;
          const unsigned cd_n = b_cdesc[ctg * 16 + n];
          const bool vec = (__builtin_amdgcn_readfirstlane(cd_n) >> 12) & 1u;
          const int xo = cd_n & 0xfff, so = (cd_n >> 16) & 15;
          const float yv = (cd_n >> 31) ? sY : 0.f;            // (padding columns: zero)
          float y[8];
          if (!vec) {                                          // x[u] sh0
            float xs[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) xs[t] = d.x[w_row[t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)] + xo];
#pragma unroll
            for (int t = 0; t < 8; ++t) y[t] = xs[t] * w_sh[(t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)) * 12 + so] * yv;
          } else {                                             // xv[u] . sh1
            float xs[8][3];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const float* xp = d.x + w_row[t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)] + xo;
              xs[t][0] = xp[0]; xs[t][1] = xp[1]; xs[t][2] = xp[2];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const float* sp = w_sh + (t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)) * 12 + so;
              y[t] = (xs[t][0] * sp[0] + xs[t][1] * sp[1] + xs[t][2] * sp[2]) * yv;
            }
          }
          u32x4 Yh, Yl;
#pragma unroll
          for (int r = 0; r < 4; ++r) { unsigned hi, lo; cz_split2(y[2 * r], y[2 * r + 1], hi, lo); Yh[r] = hi; Yl[r] = lo; }
#pragma unroll
          for (int j = 0; j < CZ_MAXSEG; ++j) {                // (segments the chunk does not have: all-zero masks)
            const u32x4 mk = *reinterpret_cast<const u32x4*>(w_mask + (j * 4 + g) * 4);
            Ym[j][0] = Yh & mk; Ym[j][1] = Yl & mk;
          }
        };
#define YM(j, p) __builtin_bit_cast(f16x8, Ym[j][p])
        // step A of one (c, k) tile: Z[c, k] of my four segment slots -- independent three-product chains side by side -- x 2^-20 (|Z| <= 32 x 2^15 x 2^15), cut
        // into two fp16 pieces and stored (column = segment): the B operand of step B as the column waves read it
        auto stepA = [&](auto j_c, int boff) {
          constexpr int jj = decltype(j_c)::value;
          if (ABL & 4) return;
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          const f16x8 hh = __builtin_bit_cast(f16x8, Hh[jj][0]), hl = __builtin_bit_cast(f16x8, Hh[jj][1]);
          f32x4 z[CZ_MAXSEG];
#pragma unroll
          for (int j = 0; j < CZ_MAXSEG; ++j) z[j] = MF(YM(j, 0), hl, zero);
#pragma unroll
          for (int j = 0; j < CZ_MAXSEG; ++j) z[j] = MF(YM(j, 1), hh, z[j]);
#pragma unroll
          for (int j = 0; j < CZ_MAXSEG; ++j) z[j] = MF(YM(j, 0), hh, z[j]);
#pragma unroll
          for (int j = 0; j < CZ_MAXSEG; ++j) {
            unsigned h0, l0, h1, l1;
            cz_split2(z[j][0] * zs, z[j][1] * zs, h0, l0);
            cz_split2(z[j][2] * zs, z[j][3] * zs, h1, l1);
            *reinterpret_cast<u32x2*>(za[j] + boff) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(za[j] + boff + CZ_PLANE) = (u32x2){l0, l1};
          }
        };
#undef YM
        const int n_ct = W.ct0[W.n_io - 1] + W.nct[W.n_io - 1];
        for (int io = 0; io < W.n_io; ++io) {
          const int ct0 = W.ct0[io];
          int tog = 0;                                         // byte offset of the buffer the running tile is READ from (step A of the next tile writes the other one)
          make_Y(ct0);
          stepA(std::integral_constant<int, 0>{}, 0);
          __syncthreads();   // B3
          for (int ct = 0; ct < W.nct[io]; ++ct) {
            cz_static_for<0, KTP>([&](auto j_c) {
              constexpr int jj = decltype(j_c)::value;
              stamp(10);
              if constexpr (jj + 1 < KTP) stepA(std::integral_constant<int, jj + 1>{}, tog ^ CZ_TOG);
              else if (ct + 1 < W.nct[io]) { make_Y(ct0 + ct + 1); stepA(std::integral_constant<int, 0>{}, tog ^ CZ_TOG); }   // the column waves are in this c tile's last k tile: the next c tile's Y and its first step A
              stamp(12);
              tog ^= CZ_TOG;
              __syncthreads();   // B4: the next tile's Z is complete; this tile's buffer may be written again
              stamp(13);
            });
          }
          __syncthreads();   // B5: the column waves' partial sums
          store_irrep(d, W, io, ncb, pass);
          __syncthreads();   // B6
        }
        (void)n_ct;
      }
      if (a.executed && lane == 0) {                           // (profiling) 16 x 16 x 32 x 2 flops per matrix instruction: the hidden layer 14 per k tile and edge tile, step A 3 per (segment slot, tile)
        long long n_mfma = 2LL * 2 * KTP * 14;
        double useful = 144.0 * 144.0 * len;
        for (int io = 0; io < W.n_io; ++io) {
          n_mfma += (long long)W.nct[io] * (3LL * CZ_NKT * CZ_MAXSEG);
          useful += (double)W.nc_valid[io] * 145.0 * ((double)len + 48.0 * nseg_u);
        }
        atomicAdd(a.executed, 16384.0 * (double)n_mfma);
        atomicAdd(a.executed + 1, 6.0 * useful);
        atomicAdd(a.executed + 2, 4.0 * 48.0 * W.n_io * nseg_u);
      }
    }
  } else {
    // =============================================================== column waves: wave NWA + b owns the k-steps KS b .. KS b + KS - 1 of every tile
    const int bw = wave - NWA;
    for (int unit = blockIdx.x; unit < N; unit += gridDim.x) {
      int c = 0, ul = unit;
      if (ul >= nu[0]) { ul -= nu[0]; c = 1; if (ul >= nu[1]) { ul -= nu[1]; c = 2; if (ul >= nu[2]) { ul -= nu[2]; c = 3; } } }
      const ConvZDesc& d = a.c[c];
      const ConvZ& W = d.w;
      tr_on = (ABL & 128) && blockIdx.x == 0 && unit == 2 * (int)gridDim.x && (a.dbg_sel == 0 || a.dbg_sel == a.c[0].w.ct0[a.c[0].w.n_io - 1] + a.c[0].w.nct[a.c[0].w.n_io - 1]);
      if (tr_on) tr_n = 0;
      stamp(1);
      int ncb = 0;
      long long n_mfma = 0;
      for (int pass = 0; pass < 2; ++pass) {
        {                                                      // this pass's half of W1h into LDS (tile jj of the pass at jj x 9280 of the Z area, idle until P2): through registers,
          // all requests of a wave in flight at once; wave b takes the pieces q = b, b + 4, b + 8 (< 10) of every tile
          const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)W.W1h, 0, KT * CH_TILE_BYTES, 0x00020000);
          u32x4 buf[KTP][3];
#pragma unroll
          for (int jj = 0; jj < KTP; ++jj)
#pragma unroll
            for (int k = 0; k < 3; ++k)                          // (beyond the buffer -- the second pass's fifth tile -- the loads return zeros)
              buf[jj][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rW1, lane * 16, (5 * pass + jj) * CH_TILE_BYTES + (bw + 4 * k) * 1024, 0));
          if (pass == 0) {                                     // the block-level tables
            const int t = tid - 64 * NWA;                      // 0 .. 255
            if (t < CZ_MAXCT * 16) b_cdesc[t] = t < (W.ct0[W.n_io - 1] + W.nct[W.n_io - 1]) * 16 ? W.cdesc[t] : 0u;
            if (t < 16 * NCB) b_col_edge[t] = -1;
          }
#pragma unroll
          for (int jj = 0; jj < KTP; ++jj)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const int q = bw + 4 * k;
              if (q < 9 || (q == 9 && lane < 4))                 // (the bias piece is 64 bytes)
                *reinterpret_cast<u32x4*>(zb + jj * CH_TILE_BYTES + q * 1024 + lane * 16) = buf[jj][k];
            }
        }
        stamp(31);
        __syncthreads();   // P1
        __syncthreads();   // P2
        stamp(33);
        if (pass == 0) {
          int total = 0;
#pragma unroll
          for (int v = 0; v < NWA; ++v) total += b_nseg[v];
          total = __builtin_amdgcn_readfirstlane(total);
          ncb = (total + 15) >> 4;
        }
        for (int io = 0; io < W.n_io; ++io) {
          // step B: 9 per (column block, k-step, tile) -- of the k tile 9 one k-step, by column wave 0 only
          n_mfma += (long long)W.nct[io] * (9LL * ncb * ((pass ? KTP - 1 : KTP) * KS + (pass && bw == 0)));
          f32x4 acc[3][NCB];
#pragma unroll
          for (int wt = 0; wt < 3; ++wt)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[wt][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const char* wbase = reinterpret_cast<const char*>(W.W2z) + (size_t)lane * 16 + (size_t)bw * (KS * CZ_TILE_BYTES);
          // W2z tile of (c tile ct, k tile 5 pass + jj): (ct0 + ct) 10 + 5 pass + jj; the running index gq walks the pass's tiles, gqn = the tile after
          auto tile_q = [&](int ct, int jj) { return (W.ct0[io] + ct) * CZ_NKT + KTP * pass + jj; };
          const int q_last = tile_q(W.nct[io] - 1, KTP - 1);
          // W2' fragments of my k-steps [k-step][w tile][hi, lo]: ONE tile in registers; a fragment pair is requested again, for the next tile, as soon as its
          // last matrix instruction has been issued (what bounds the tile loop is the stream of these fragments out of L2 at ~26 B per cycle and workgroup,
          // not its latency: profiles/TUNING_r6.md)
          u32x4 Wf[KS][3][2];
          auto fetchG = [&](auto ks_c, auto wt_c, int q) {
            constexpr int ks = decltype(ks_c)::value, wt = decltype(wt_c)::value;
            const char* p = wbase + (size_t)q * (8 * CZ_TILE_BYTES) + ks * CZ_TILE_BYTES + wt * 2048;
            Wf[ks][wt][0] = *reinterpret_cast<const u32x4*>(p);
            Wf[ks][wt][1] = *reinterpret_cast<const u32x4*>(p + 1024);
          };
          int ct = 0;
          auto next_q = [&](int jj) { return jj + 1 < KTP ? tile_q(ct, jj + 1) : min(tile_q(ct + 1, 0), q_last); };   // (no branch around the fetch: the last one fetches a tile again)
          // One tile: the Z pieces of my k-steps, then the KS x 3 fragment pairs two at a time -- their 6 NCBV matrix instructions interleaved (W_hi Z_lo,
          // W_lo Z_hi, W_hi Z_hi per column block) -- and behind each two pairs their successors of the next tile
          auto tileB = [&](auto j_c, auto ncb_c, int tog) {
            constexpr int jj = decltype(j_c)::value, NCBV = decltype(ncb_c)::value;
            const char* zr = zb + tog + (bw * KS) * CZ_VSTRIDE + g * 256 + n * 16;
            const int qn = next_q(jj);
            if (jj == KTP - 1 && pass) {
              // the k tile 9 (the bias row: k-step 0 only, column wave 0's first); every wave's fragments of this tile make way for those of the next
              if (bw == 0 && !(ABL & 8)) {
#pragma unroll
                for (int cb = 0; cb < NCBV; ++cb) {
                  const f16x8 zh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(zr + cb * CZ_CB));
                  const f16x8 zl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(zr + cb * CZ_CB + CZ_PLANE));
#pragma unroll
                  for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[0][wt][0]), zl, acc[wt][cb]);
#pragma unroll
                  for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[0][wt][1]), zh, acc[wt][cb]);
#pragma unroll
                  for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[0][wt][0]), zh, acc[wt][cb]);
                }
              }
              SL;
              cz_static_for<0, KS * 3>([&](auto f_c) { constexpr int f = decltype(f_c)::value; fetchG(std::integral_constant<int, f / 3>{}, std::integral_constant<int, f % 3>{}, qn); });
              return;
            }
            // k-step by k-step, column block by column block: the block's pieces (read one block ahead), 9 matrix instructions (three accumulators in turn);
            // behind a k-step's last block its three fragment pairs of the next tile
            u32x4 ph[2], pl[2];
            SL;
            ph[0] = *reinterpret_cast<const u32x4*>(zr);
            pl[0] = *reinterpret_cast<const u32x4*>(zr + CZ_PLANE);
            cz_static_for<0, KS * NCBV>([&](auto b_c) {
              constexpr int b = decltype(b_c)::value, ks = b / NCBV, cb = b % NCBV, cur = b & 1;
              SL;
              if constexpr (b + 1 < KS * NCBV) {
                constexpr int ks2 = (b + 1) / NCBV, cb2 = (b + 1) % NCBV;
                ph[cur ^ 1] = *reinterpret_cast<const u32x4*>(zr + ks2 * CZ_VSTRIDE + cb2 * CZ_CB);
                pl[cur ^ 1] = *reinterpret_cast<const u32x4*>(zr + ks2 * CZ_VSTRIDE + cb2 * CZ_CB + CZ_PLANE);
              }
              if constexpr ((ABL & 8) == 0) {
                const f16x8 zh = __builtin_bit_cast(f16x8, ph[cur]), zl = __builtin_bit_cast(f16x8, pl[cur]);
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[ks][wt][0]), zl, acc[wt][cb]);
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[ks][wt][1]), zh, acc[wt][cb]);
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[ks][wt][0]), zh, acc[wt][cb]);
              } else {
                asm volatile("" :: "v"(ph[cur]), "v"(pl[cur]));
              }
              if constexpr (cb == NCBV - 1) {
                SL;
                cz_static_for<0, 3>([&](auto wt_c) { fetchG(std::integral_constant<int, ks>{}, wt_c, qn); });
              }
              SL;
            });
          };
          cz_static_for<0, KS * 3>([&](auto f_c) { constexpr int f = decltype(f_c)::value; fetchG(std::integral_constant<int, f / 3>{}, std::integral_constant<int, f % 3>{}, tile_q(0, 0)); });
          __syncthreads();   // B3
          // (the column-block count is decided once per unit, OUTSIDE the tile loops)
          auto run_tiles = [&](auto ncb_c) {
            int tog = 0;
            for (ct = 0; ct < W.nct[io]; ++ct) {
              cz_static_for<0, KTP>([&](auto j_c) {
                stamp(10);
                tileB(j_c, ncb_c, tog);
                stamp(12);
                tog ^= CZ_TOG;
                __syncthreads();   // B4
                stamp(13);
              });
            }
          };
          if (ncb <= 1) run_tiles(std::integral_constant<int, 1>{});
          else if (ncb == 2) run_tiles(std::integral_constant<int, 2>{});
          else run_tiles(std::integral_constant<int, 3>{});
          // ---- the output irrep is complete for this pass: my partial sums (over my k-steps) to LDS; all sixteen waves add and store
          float* red = reinterpret_cast<float*>(zb) + bw * (12 * NCB * 64);
#pragma unroll
          for (int wt = 0; wt < 3; ++wt)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
              for (int q = 0; q < 4; ++q) red[((wt * NCB + cb) * 4 + q) * 64 + lane] = acc[wt][cb][q];
          __syncthreads();   // B5
          store_irrep(d, W, io, ncb, pass);
          __syncthreads();   // B6
        }
      }
      if (a.executed && lane == 0) atomicAdd(a.executed, 16384.0 * (double)n_mfma);
    }
  }
}
#undef MF
#undef SL

size_t convz_lds_bytes() { return CZ_ZBYTES + CZ_NWA * CZ_WAVE_FLOATS * sizeof(float) + CZ_BLOCK_INTS * sizeof(int); }

void launch_convz(const ConvZArgs& a0, hipStream_t st) {
  ConvZArgs a = a0;
  static float* dbg_dev = nullptr;
  static const char* dbg_file = getenv("DBFR_CONVZ_DEBUG");
  if (dbg_file && !dbg_dev) {
    if (hipMalloc(&dbg_dev, 16 * 512 * 8) != hipSuccess || hipMemset(dbg_dev, 0, 16 * 512 * 8) != hipSuccess) dbg_dev = nullptr;
    else atexit([] {
      std::vector<float> h(16 * 512 * 2);
      if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h.data(), dbg_dev, h.size() * 4, hipMemcpyDeviceToHost) == hipSuccess)
        if (FILE* f = fopen(getenv("DBFR_CONVZ_DEBUG"), "wb")) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
    });
  }
  a.dbg = dbg_dev;
  static const int dbg_sel = getenv("DBFR_CONVZ_DEBUG_SEL") ? atoi(getenv("DBFR_CONVZ_DEBUG_SEL")) : 0;
  a.dbg_sel = dbg_sel;
  const size_t lds = convz_lds_bytes();
#define V(x) { if (dbfr_launch_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_convz<x>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "k_convz: hipFuncSetAttribute(MaxDynamicSharedMemorySize)")) return; \
               hipLaunchKernelGGL((k_convz<x>), dim3(dbfr_current_cu_count()), dim3(64 * CZ_NW), lds, st, a); return; }
#ifdef DBFR_DEV_VARIANTS
  static int abl = getenv("DBFR_CONVZ_ABL") ? atoi(getenv("DBFR_CONVZ_ABL")) : 0;
  if (abl == 4) V(4) if (abl == 8) V(8) if (abl == 12) V(12) if (abl == 128) V(128)
#endif
  V(0)
#undef V
}
