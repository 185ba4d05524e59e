// Reduce-first form of the SCALAR-OUTPUT tensor-product paths of a K = 144 conv (DBFR_GEMM_REDUCE_FIRST; replaces
// tpscore.py:177-199 for the rows of lin.3 that feed an l = 0 output irrep -- 74 % of W at depth 3, all of W in the torsion convs).
//
// What the reference does per edge:  w[e] = W2 h[e] + b2  (144 x W),  m[e, o] = sum_u w[e, (p,u,o)] y_p[e,u]  with y_p[e,u] = x[gth e, u] sh0[e]
// (scalar input) or xv[gth e, u] . sh1[e] (vector input), then out[t] = mean over the edges of target t.  For a scalar output the whole
// chain is linear in y (x) h, so the order can be turned round:
//     Z[t, c, k] = sum_{e -> t} y[e, c] h'[e, k]          (c = (path, u) pairs into the irrep, h' = [h | 1]: the 1 carries b2)
//     out_sum[t, o] = sum_{c,k} Z[t, c, k] W2'[c, k, o]   (W2' = lin.3 rows x the folded path constant)
// -- the 144 x W GEMM once per TARGET (segment of <= 32 edges), not once per edge: 8-10 x fewer matrix instructions at the 13-50 edges
// per node of the pocket / ligand graphs.  The l = 1 outputs stay per edge (k_conv2h on a W2 without these rows): for them Z is three
// times as large and the saving is nil.
//
// Both products run on v_mfma_f32_16x16x32_f16 with fp32 operands cut into two fp16 pieces / three partial products, as in conv2h.hip:
//   * a wave owns a CHUNK: consecutive edges of one graph, at most 32 of them and at most CZ_MAXSEG = 4 targets (the chunk table of graph.hip
//     k_chunk_count / k_graph_chunks / k_chunk_fill; chunks are cut per graph by the graph's own targets, so what is summed with what never depends
//     on batch mates); the maximal runs of one target inside the chunk are its SEGMENTS.  Hidden layer transposed, D[edge, unit] = A W1^T: the
//     radial-MLP inputs are the A operand straight from memory, the W1h tiles of conv2h serve unchanged as B operand, and the result registers --
//     unit on the lane, eight edges in registers -- ARE the B operand of step A (contraction over the edges);
//   * Y of a c tile (gathered x rows x harmonics, cut into pieces) is masked per segment ONCE per c tile and kept in registers (Ym[4][hi, lo]);
//   * a tile (c, k) is one instruction stream per wave: step A of the NEXT tile -- four independent three-product chains Ym[j] x H[k + 1], the
//     16 x 16 blocks of Z as fp32 into the other LDS buffer, column = segment -- with the cutting of this tile's Z in its shadow, then step B of this
//     tile: the eight waves take one k-step of 32 each of the 256 (c, k) values of every column, W2' fragments straight from L2 into registers a
//     whole tile ahead (every wave another k-step: no LDS ring), columns = the up to 32 segments of the workgroup's eight chunks; partial sums over
//     the k-steps are added across the waves once per output irrep.
// Scaling (exact powers of two): inputs per edge, W1 per matrix or per row (conv2h), h per chunk, y per chunk (bound from max |x| max |sh|), Z by the
// constant 2^-20 (|Z| <= 32 x 2^15 x 2^15), W2' per output ROW (undone on the accumulator rows at the end: no row-depth limit).
// Message interface: the sum of a segment lands in the message row of the segment's FIRST edge; the scalar columns of its other rows are not
// written -- k_reduce_ln[_layer] add a node's flagged rows for these columns (EdgeSet::seg_first, written with the chunk table) and all rows
// for the vector columns, and divide by the number of edges as before.
// Design, cost model and what bounds the kernel: docs/kernels/conv_reduce_first.md; what was tried: profiles/TUNING_r5.md.
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

#define CH_TILE_BYTES 9280          // W1h tile format of conv2h.hip
#define CH_TAIL_OFF 8192
#define CH_BIAS_OFF 9216

// Z in LDS (fp32, as step A's accumulators leave it), per buffer and column block: [k-step v: 2064 B][lane group g'': 512 B][half h: 256 B][column: 16 B]
// = 4 values.  Step A's lane (k_local n, group g) writes its four values c_local = 4 g + q as ONE 16-byte store to (v = n & 7, g'' = g,
// h = n >> 3); step B's lane (column, g'') reads the two halves of its k-step and cuts them into fp16 pieces itself -- the cutting is spread
// evenly over the eight waves instead of sitting on the wave with the most segments.  The 2064-byte stride (4 dwords mod 32 banks) makes
// the stores conflict-free, the 256-byte halves / 512-byte groups (0 mod 64 banks) the loads.
#define CZ_VSTRIDE 2064
#define CZ_CB (8 * CZ_VSTRIDE)      // one column block
#define CZ_NCB 2                    // column blocks of 16: the <= 8 x CZ_MAXSEG segments of the workgroup's eight chunks take consecutive columns
#define CZ_TILE (CZ_NCB * CZ_CB)    // one (c, k) tile: all column blocks
#define CZ_KPB 1                    // k tiles per barrier
#define CZ_BUF (CZ_KPB * CZ_TILE)
#define CZ_ZBYTES (2 * CZ_BUF)      // two buffers: step A of the next tile writes while step B of this one reads
#define CZ_W1BYTES (9 * CH_TILE_BYTES)   // the conv's W1h tiles, copied into the Z region (+ 17 KB behind it) for the hidden layer of every unit: Z is idle until the first tile
#define CZ_R0BYTES (CZ_W1BYTES > CZ_ZBYTES ? CZ_W1BYTES : CZ_ZBYTES)
#define CZ_WAVE_FLOATS (32 * 12 + 32 + 32 + 32 + 32 + 32 + CZ_MAXSEG * 16 + 24 * 64)   // harmonics [32][12] | sa | ua | gather row offsets | segment ids | first slot of segment j | masks [CZ_MAXSEG segments][4 lane groups][4 dwords] | gathered x of the next c tile [8 slots x 3 components][64 lanes]
#ifndef CZ_PRIO
#define CZ_PRIO 1
#endif
#if CZ_PRIO
#define CZ_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define CZ_SETPRIO(x)
#endif
#ifndef CZ_FETCH_AT
#define CZ_FETCH_AT 4                // step A's matrix instructions in front of the next tile's W2' fetch (0 | 4 | 8 | 12)
#endif
#ifndef CZ_PF_KT
#define CZ_PF_KT 7                  // the k tile behind which the next c tile's x gathers set out (used at k tile 9, behind an explicit wait); 3 / 5 / 6 / 8: 2.5-3.5 % slower (same-box A/B, profiles/TUNING_r5.md)
#endif
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

template <int NW, int ABL = 0>   // ABL (developer, wrong results): 1 no W2' fetch beyond the first tile, 2 no barriers in the tile loop, 4 no step A, 8 no step B, 16 no y gathers, 32 prologue only
__global__ __launch_bounds__(64 * NW, 2) void k_convz(ConvZArgs a) {
  constexpr int KT = 9;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave: a scalar register)
  const int n = lane & 15, g = lane >> 4;
  char* zb = reinterpret_cast<char*>(lds);
  float* wl = lds + CZ_R0BYTES / 4 + wave * CZ_WAVE_FLOATS;
  float* w_sh = wl;                                         // [32][12]
  float* w_sa = wl + 32 * 12;                               // [32] the edge's factor on its radial-MLP inputs
  float* w_ua = w_sa + 32;                                  // [32] ... and its inverse
  int* w_row = reinterpret_cast<int*>(w_ua + 32);           // [32] gth[e] * ldx
  int* w_seg = w_row + 32;                                  // [32] segment of the slot, -1: no edge
  int* w_first = w_seg + 32;                                // [32] first slot of segment j
  unsigned* w_mask = reinterpret_cast<unsigned*>(w_first + 32);   // [CZ_MAXSEG][4][4] segment j's mask on the A-operand registers of lane group g
  float* w_xs = reinterpret_cast<float*>(w_mask + CZ_MAXSEG * 16);   // [24][64] the next c tile's gathered x values: row = slot t (+ 8 m: component m of a vector input), one dword per lane
  int* b_nseg = reinterpret_cast<int*>(lds + CZ_R0BYTES / 4 + NW * CZ_WAVE_FLOATS);   // [NW]
  int* b_col_edge = b_nseg + NW;                            // [48] message row of the column's segment, -1: column unused
  float* b_col_inv = reinterpret_cast<float*>(b_col_edge + 16 * CZ_NCB);   // [32] takes the chunk's factors off
  unsigned* b_cdesc = reinterpret_cast<unsigned*>(b_col_inv + 16 * CZ_NCB);   // [CZ_MAXCT][16] the conv's c-tile descriptors (a dependent global load per use otherwise)

  // developer timeline (ABL & 128, DBFR_CONVZ_DEBUG=<file>): s_memtime stamps of workgroup 0's first unit, [wave][stamp] unsigned long long in a.dbg
  int tr_n = 0;
  auto stamp = [&](int tag) {
    if constexpr ((ABL & 128) != 0) {
      if (a.dbg && blockIdx.x == 0 && lane == 0 && tr_n < 1000) {
        unsigned long long* t = reinterpret_cast<unsigned long long*>(a.dbg) + wave * 1024;
        t[tr_n++] = (__builtin_amdgcn_s_memtime() << 8) | (unsigned)tag;
      }
    }
  };
  // ---- unit list: NW chunks of 32 edges per unit, conv after conv
  int nch[4] = {0, 0, 0, 0}, nu[4] = {0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < a.n_conv) {
      const ConvZDesc& d = a.c[c];
      nch[c] = min(*d.n_chunks, d.max_chunks);
      nu[c] = (nch[c] + NW - 1) / NW;
    }
  const int N = nu[0] + nu[1] + nu[2] + nu[3];
  for (int unit = blockIdx.x; unit < N; unit += gridDim.x) {
    int c = 0, ul = unit;
    if (ul >= nu[0]) { ul -= nu[0]; c = 1; if (ul >= nu[1]) { ul -= nu[1]; c = 2; if (ul >= nu[2]) { ul -= nu[2]; c = 3; } } }
    const ConvZDesc& d = a.c[c];
    const ConvZ& W = d.w;
    const int E = min(*d.n_edges, d.max_edges);
    // ---- my chunk: edges [es, es + len)
    const int ch = ul * NW + wave;
    int es = 0, len = 0, gidx = 0;
    if (ch < nch[c]) {                                       // (k_chunk_fill: <= 32 edges and <= CZ_MAXSEG targets of one graph)
      es = d.chunk_es[ch];
      const int gl = d.chunk_gl[ch];
      gidx = gl >> 6;
      len = min(min(gl & 63, 32), max(E - es, 0));
    }
    __syncthreads();   // nothing of the last unit still reads the block-level arrays
    if (tid < CZ_MAXCT * 16) b_cdesc[tid] = tid < (W.ct0[W.n_io - 1] + W.nct[W.n_io - 1]) * 16 ? W.cdesc[tid] : 0u;
    if (unit != (int)blockIdx.x) tr_n = 1000;                // (the first unit only)
    // W1h -> LDS, a 1 / (64 NW)-th per thread: requested here, written in front of the hidden layer (the eight waves used to fetch the same 83 KB
    // from L1 / L2 each, two steps ahead of their MFMAs: 45 waits of an L1 round trip per chunk)
    constexpr int W1_V4 = CZ_W1BYTES / 16, W1_IT = (W1_V4 + 64 * NW - 1) / (64 * NW);
    u32x4 stg[W1_IT];
#pragma unroll
    for (int i = 0; i < W1_IT; ++i) {
      const int j = tid + i * 64 * NW;
      stg[i] = reinterpret_cast<const u32x4*>(W.W1h)[min(j, W1_V4 - 1)];
    }
    stamp(1);
    // ---- slots: lanes 0..31 own slot L (clamped to the chunk's last edge beyond its length; a chunk without edges reads edge 0 of a non-empty conv or nothing)
    const int sl = lane & 31;
    const bool have = len > 0;
    const int e_sl = have ? es + min(sl, len - 1) : 0;
    int tgt_l = 0, gth_l = 0;
    if (have) { tgt_l = d.tgt[e_sl]; gth_l = d.gth[e_sl]; }
    const int tgt_prev = __shfl_up(tgt_l, 1);
    const bool is_first = lane < 32 && sl < len && (sl == 0 || tgt_l != tgt_prev);
    const unsigned long long bal = __ballot(is_first);
    const unsigned firsts = (unsigned)bal;
    const int nseg = min(__popc(firsts), CZ_MAXSEG);         // (the chunk table holds no chunk with more)
    const int seg_l = (sl < len) ? __popc(firsts & (0xffffffffu >> (31 - sl))) - 1 : -1;
    if (lane < 32) {
      w_row[sl] = gth_l * d.ldx;
      w_seg[sl] = seg_l;
      if (is_first) w_first[seg_l] = sl;
      const float* sp = d.sh + (size_t)e_sl * SH_LD;
#pragma unroll
      for (int k = 0; k < 9; ++k) w_sh[sl * 12 + k] = have ? sp[k] : 0.f;
      w_sh[sl * 12 + 9] = w_sh[sl * 12 + 10] = w_sh[sl * 12 + 11] = 0.f;
    }
    if (lane == 0) b_nseg[wave] = nseg;
    // (the scalar columns of the message rows that are not the first of their segment are NOT written: the reductions read those columns of flagged rows
    // only -- EdgeSet::seg_first, conv.hip row_sum; round 5's first kernel zeroed them, 1.5 GB of stores per layer launch, and the reduction read them back)
    // ---- bounds for the y scale: largest |x| over the gathered rows, largest |harmonic|
    float xmx = 0.f, smx = 0.f;
    if (have) {
      if (d.xmax) xmx = d.xmax[gidx];                       // per graph, computed once per layer (k_row_absmax): no pass over the gathered rows here
      else {
        const float* xr = d.x + (size_t)gth_l * d.ldx;
        const int d4 = d.ldx >> 2;
        for (int j = lane >> 5; j < d4; j += 2) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * j);
          xmx = fmaxf(fmaxf(xmx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
      }
      if (lane < 32) {
        const float* sp = d.sh + (size_t)e_sl * SH_LD;
#pragma unroll
        for (int k = 0; k < 9; ++k) smx = fmaxf(smx, fabsf(sp[k]));
      }
    }
    xmx = cz_wave_max(xmx); smx = cz_wave_max(smx);
    int ey = 0;                                              // y is multiplied by 2^ey: |y| <= 3 |x| |sh| < 2^(ex + es + 2) -> below 2^15
    if (xmx > 0.f && smx > 0.f) ey = max(-100, min(100, 13 - __builtin_amdgcn_frexp_expf(xmx) - __builtin_amdgcn_frexp_expf(smx)));
    ey = __builtin_amdgcn_readfirstlane(ey);
    const float sY = __builtin_amdgcn_ldexpf(1.f, ey);

    stamp(2);
    // ---- hidden layer, transposed: D[edge, unit] = sum_f a[edge, f] W1[unit, f]; A = the edge's inputs (cut per edge), B = W1h tiles
    u32x4 Hh[CZ_NKT][2];                                     // H pieces [k tile][hi, lo]: lane (unit n, group g), eight edges {4g..4g+3, 16+4g..16+4g+3}
    int eh = 0;                                              // the factor on h is 2^(15 - ehc + k1) = phi, also what the constant 1 of the bias becomes
    {
      u32x4 Ah[2][2][4];                                     // input pieces [edge tile][hi, lo][k-step of 32]
      u32x4 Atc[2];                                          // last 16 k: [lo | hi]
#pragma unroll
      for (int et = 0; et < 2; ++et) {
        const int slot = 16 * et + n;
        const int e = have ? es + min(slot, len - 1) : 0;
        const float* r0 = d.emb + (size_t)e * NS;
        const float* r1 = d.tab1 + (size_t)(have ? d.idx1[e] : 0) * d.ld1;
        const float* r2 = d.tab2 + (size_t)(have ? d.idx2[e] : 0) * d.ld2;
        f32x4 Ba[KT];
        float amx = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < KT; ++s4) {
          const float* src = s4 < 3 ? r0 : s4 < 6 ? r1 : r2;
          Ba[s4] = have ? *reinterpret_cast<const f32x4*>(src + 16 * (s4 % 3) + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) amx = fmaxf(amx, fabsf(Ba[s4][q]));
        }
        amx = fmaxf(amx, __shfl_xor(amx, 16));
        amx = fmaxf(amx, __shfl_xor(amx, 32));
        const int ja = max(-14, __builtin_amdgcn_frexp_expf(amx));
        const float sa = __builtin_amdgcn_ldexpf(1.f, 15 - ja);
        if (g == 0) { w_sa[slot] = sa; w_ua[slot] = __builtin_amdgcn_ldexpf(1.f, ja - 15); }
#pragma unroll
        for (int s4 = 0; s4 < KT; ++s4) {
          unsigned hi0, lo0, hi1, lo1;
          cz_split2(Ba[s4][0] * sa, Ba[s4][1] * sa, hi0, lo0);
          cz_split2(Ba[s4][2] * sa, Ba[s4][3] * sa, hi1, lo1);
          if (s4 < 8) {
            Ah[et][0][s4 >> 1][2 * (s4 & 1)] = hi0; Ah[et][0][s4 >> 1][2 * (s4 & 1) + 1] = hi1;
            Ah[et][1][s4 >> 1][2 * (s4 & 1)] = lo0; Ah[et][1][s4 >> 1][2 * (s4 & 1) + 1] = lo1;
          } else {
            Atc[et] = (u32x4){lo0, lo1, hi0, hi1};
          }
        }
      }
      stamp(4);
#pragma unroll
      for (int i = 0; i < W1_IT; ++i) {
        const int j = tid + i * 64 * NW;
        if (j < W1_V4) reinterpret_cast<u32x4*>(zb)[j] = stg[i];
      }
      __syncthreads();
      stamp(5);
      // my rows of a result tile are the edges 16 et + 4 g + q: their factors
      f32x4 sar[2], uar[2];
#pragma unroll
      for (int et = 0; et < 2; ++et) {
        sar[et] = *reinterpret_cast<const f32x4*>(w_sa + 16 * et + 4 * g);
        uar[et] = *reinterpret_cast<const f32x4*>(w_ua + 16 * et + 4 * g);
      }
      float Hf[KT][2][4];
      float hmx = 0.f;
      const int vW = lane * 16;
      // W1 fragments: step i = 5 m + s (s < 4: the k-step of 32 of tile m, hi and lo piece; s = 4: the tile's last 16 k [hi | lo] and, in the second
      // register, my unit's bias) through a ring of three register pairs, requested TWO steps ahead of their MFMAs (a load next to its use waits
      // out the L1 / L2 latency 45 times per chunk: 59 k of the unit's 270 k cycles in the timeline of round 5)
      u32x4 F[3][2];
      auto ldF = [&](auto ic) {
        constexpr int i = decltype(ic)::value, m = i / 5, sx = i % 5;
        if constexpr (sx < 4) {
          F[i % 3][0] = *reinterpret_cast<const u32x4*>(zb + vW + m * CH_TILE_BYTES + sx * 1024);
          F[i % 3][1] = *reinterpret_cast<const u32x4*>(zb + vW + m * CH_TILE_BYTES + (4 + sx) * 1024);
        } else {
          F[i % 3][0] = *reinterpret_cast<const u32x4*>(zb + vW + m * CH_TILE_BYTES + CH_TAIL_OFF);
          F[i % 3][1][0] = *reinterpret_cast<const unsigned*>(zb + n * 4 + m * CH_TILE_BYTES + CH_BIAS_OFF);
        }
      };
      ldF(std::integral_constant<int, 0>{});
      ldF(std::integral_constant<int, 1>{});
      // (tile m's bias arrives with its LAST step; the accumulators start from zero and the bias x the edge's factor is added behind the tile)
      f32x4 acc[2];
      cz_static_for<0, 5 * KT>([&](auto ic) {
        constexpr int i = decltype(ic)::value, m = i / 5, sx = i % 5;
        if constexpr (i + 2 < 5 * KT) ldF(std::integral_constant<int, (i + 2 < 5 * KT ? i + 2 : 0)>{});
        if constexpr (sx == 0) { acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        if constexpr (sx < 4) {
          const f16x8 whi = __builtin_bit_cast(f16x8, F[i % 3][0]), wlo = __builtin_bit_cast(f16x8, F[i % 3][1]);
#pragma unroll
          for (int et = 0; et < 2; ++et) {
            acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Ah[et][1][sx]), whi, acc[et], 0, 0, 0);
            acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Ah[et][0][sx]), wlo, acc[et], 0, 0, 0);
            acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Ah[et][0][sx]), whi, acc[et], 0, 0, 0);
          }
        } else {
          // k = 128..143: W tail fragment = [hi (4) | lo (4)], input tail = [lo | hi]: one x32 MFMA carries both small products; the large product
          // on the x32 instruction too, upper half of the operands zero (an x16 MFMA that takes as SrcC an accumulator an x32 MFMA has just
          // written reads stale data on MI355X + ROCm 7.2 -- conv2r.hip -- and 18 half-empty MFMAs per chunk cost nothing)
          const u32x4 wt = F[i % 3][0];
          const f16x8 wf = __builtin_bit_cast(f16x8, wt);
          const f16x8 wh = __builtin_bit_cast(f16x8, (u32x4){wt[0], wt[1], 0u, 0u});
          const float bias = __builtin_bit_cast(float, F[i % 3][1][0]);
          const float w1r = W.W1rinv ? W.W1rinv[16 * m + n] : 1.f;
#pragma unroll
          for (int et = 0; et < 2; ++et) {
            acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Atc[et]), wf, acc[et], 0, 0, 0);
            acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, (u32x4){Atc[et][2], Atc[et][3], 0u, 0u}), wh, acc[et], 0, 0, 0);
          }
#pragma unroll
          for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float v = fmaxf(acc[et][q] + bias * sar[et][q], 0.f) * (uar[et][q] * w1r);   // = 2^k1 h (w1r: my unit's row factor of W1h off)
              Hf[m][et][q] = v;
              hmx = fmaxf(hmx, v);
            }
        }
      });
      if (!(ABL & 128) && a.dbg && unit == 0 && wave == 0) {
        const float k1i = __builtin_amdgcn_ldexpf(1.f, -W.k1);
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
          for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int q = 0; q < 4; ++q) a.dbg[(16 * et + 4 * g + q) * 144 + 16 * m + n] = Hf[m][et][q] * k1i;
      }
      stamp(6);
      hmx = cz_wave_max(hmx);
      // 2^k1 h < 2^eh; the factor 2^(15 - ehc) on it makes phi = 2^(15 - ehc + k1) on h: ehc >= k1 keeps phi, the image of the bias's 1, inside fp16
      eh = max(hmx > 0.f ? __builtin_amdgcn_frexp_expf(hmx) : W.k1, W.k1);
      eh = min(eh, W.k1 + 100);
      const float sH = __builtin_amdgcn_ldexpf(1.f, 15 - eh);
#pragma unroll
      for (int m = 0; m < KT; ++m) {
        unsigned h0, l0, h1, l1, h2, l2, h3, l3;
        cz_split2(Hf[m][0][0] * sH, Hf[m][0][1] * sH, h0, l0);
        cz_split2(Hf[m][0][2] * sH, Hf[m][0][3] * sH, h1, l1);
        cz_split2(Hf[m][1][0] * sH, Hf[m][1][1] * sH, h2, l2);
        cz_split2(Hf[m][1][2] * sH, Hf[m][1][3] * sH, h3, l3);
        Hh[m][0] = (u32x4){h0, h1, h2, h3};
        Hh[m][1] = (u32x4){l0, l1, l2, l3};
      }
    }
    const int ephi = 15 - eh + W.k1;                         // log2 of the factor on h
    {   // k tile 9: the constant 1 (x phi) of the bias in column 0, for the slots that hold an edge
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
      Hh[KT][0] = hb;
      Hh[KT][1] = (u32x4){0u, 0u, 0u, 0u};
    }
    stamp(3);
    // ---- masks of my segments on the registers of the A operand (halves of a dword = two consecutive slots), for every lane group
    {                                                        // (segments the chunk does not have: all-zero masks)
      const int j = lane >> 4, gg = (lane >> 2) & 3, r = lane & 3, t0 = 2 * r;
      const int s0 = t0 < 4 ? 4 * gg + t0 : 16 + 4 * gg + (t0 - 4);
      w_mask[lane] = (w_seg[s0] == j ? 0x0000ffffu : 0u) | (w_seg[s0 + 1] == j ? 0xffff0000u : 0u);
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    // ---- columns: the segments of the eight chunks side by side (<= 8 x CZ_MAXSEG = 32 = two column blocks of step B)
    int cbase = 0, total = 0;
#pragma unroll
    for (int v = 0; v < NW; ++v) { const int c = b_nseg[v]; if (v < wave) cbase += c; total += c; }
    cbase = __builtin_amdgcn_readfirstlane(cbase); total = __builtin_amdgcn_readfirstlane(total);   // (wave-uniform: scalar branches below)
    const int nseg_u = __builtin_amdgcn_readfirstlane(nseg);
    const int ncb = (total + 15) >> 4;                       // column blocks of 16 in step B: 0 (no edges), 1 or 2
    const float col_inv = __builtin_amdgcn_ldexpf(1.f, -CZ_ZSCALE - ey - ephi);
    const float zs = __builtin_amdgcn_ldexpf(1.f, CZ_ZSCALE);
    if (ABL & 32) continue;
    for (int c = tid; c < 16 * CZ_NCB; c += 64 * NW) b_col_edge[c] = -1;
    __syncthreads();
    if (lane < nseg_u) {
      b_col_edge[cbase + lane] = es + w_first[lane];
      b_col_inv[cbase + lane] = col_inv;
    }
    // LDS addresses of my segments' Z columns in buffer 0 (lane part + column).  A slot without a segment computes zeros and writes them into
    // column 31: unused whenever such a slot exists (the unit then has fewer than 32 segments), and never stored
    char* za[CZ_MAXSEG];
    {
      char* zw0 = zb + (n & 7) * CZ_VSTRIDE + g * 512 + (n >> 3) * 256;
#pragma unroll
      for (int j = 0; j < CZ_MAXSEG; ++j) {
        const int col = cbase + j;
        za[j] = j < nseg_u ? zw0 + (col >> 4) * CZ_CB + (col & 15) * 16 : zw0 + CZ_CB + 15 * 16;
      }
    }
    long long n_mfma = 2 * KT * 14;                          // (profiling) matrix instructions this wave issues in this unit: the hidden layer ...
    for (int io = 0; io < W.n_io; ++io) {
      // ... step A: 3 per (segment slot, tile), step B: 9 per (column block, tile) -- of the k tile 9 by wave 0 only
      n_mfma += (long long)W.nct[io] * (3LL * CZ_NKT * CZ_MAXSEG + 9LL * ncb * (KT + (wave == 0)));
      f32x4 acc[3][CZ_NCB];
#pragma unroll
      for (int wt = 0; wt < 3; ++wt)
#pragma unroll
        for (int cb = 0; cb < CZ_NCB; ++cb) acc[wt][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const char* wbase = reinterpret_cast<const char*>(W.W2z) + (size_t)lane * 16;
      int gq = W.ct0[io] * CZ_NKT;                           // running (c tile, k tile) index into W2z; CZ_NKT is even: the Z buffer of tile gq is kt & 1
      const int gq_last = (W.ct0[io] + W.nct[io]) * CZ_NKT - 1;
      u32x4 Wf[2][3][2];                                     // W2' fragments of my k-step: [tile parity][w tile][hi, lo], fetched one whole tile ahead
      auto fetchW = [&](auto par_c, int q) {
        constexpr int par = decltype(par_c)::value;
        if ((ABL & 1) && q != W.ct0[io] * CZ_NKT) return;
        // (no branch around these loads, not even for the k tile 9 whose fragments only wave 0 uses: behind a conditional fetch hipcc's wait-count
        // pass assumes the loads were NOT issued and makes step B wait for vmcnt(0), i.e. for the fragments requested a moment ago)
        const char* p = wbase + ((size_t)q * 8 + wave) * CZ_TILE_BYTES;
#pragma unroll
        for (int wt = 0; wt < 3; ++wt)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) Wf[par][wt][pc] = *reinterpret_cast<const u32x4*>(p + (wt * 2 + pc) * 1024);
      };
      // Y of a c tile, MASKED per segment: lane (c = n, group g), my eight slots; Ym[j] = the pieces with the slots outside segment j zeroed -- the
      // A operand of step A for all ten k tiles of the c tile (masking per (c, k) tile cost 8 vector instructions and a mask read per segment and
      // tile).  The gathers of a scalar-input tile are requested several tiles before they are used (load_x -> finish_Y); the one vector-input
      // tile per irrep is gathered where it is needed.
      u32x4 Ym[CZ_MAXSEG][2];
#pragma unroll
      for (int j = 0; j < CZ_MAXSEG; ++j) { Ym[j][0] = (u32x4){0u, 0u, 0u, 0u}; Ym[j][1] = (u32x4){0u, 0u, 0u, 0u}; }
      // Y of a c tile (lane: c = n, group g; my eight slots): x values gathered through the slots' row offsets x the harmonics x the chunk's power of two,
      // cut into pieces and masked per segment.  The gathers go STRAIGHT TO LDS (global_load_lds_dword: no register holds them -- under the tile loop's
      // register pressure prefetched values were spilled one by one, a load, a wait and a scratch store each), a few tiles before the c tile starts:
      // row t (+ 8 m for component m of a vector input) of the wave's stage, one dword per lane.
      // (prefetch_x / finish_Y run once per c tile: they take the lane id from v_mbcnt instead of the kernel's `lane`, so that nothing of theirs is kept
      // in registers -- or spilled and reloaded, a memory round trip each -- across the ten tiles between two uses)
      auto prefetch_x = [&](int ct) {
        if (nseg_u == 0 || (ABL & 16)) return;
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), n = lane & 15, g = lane >> 4;
        const unsigned cd_n = b_cdesc[(W.ct0[io] + ct) * 16 + n];
        const bool vec = (__builtin_amdgcn_readfirstlane(cd_n) >> 12) & 1u;
        const int xo = cd_n & 0xfff;
        // (inline assembly on purpose: for the builtin hipcc's wait-count pass makes EVERY later LDS read -- the Z reads of the next tiles -- wait for
        // vmcnt(0), i.e. for W2' fragments requested a moment before; here the one wait sits in front of the stage's readers, wait_stage)
        // (m0 is written without being declared clobbered: hipcc treats it as a reserved register and rejects the clobber; nothing the compiler emits for
        // this kernel uses m0 -- no movrel indexing, no LDS-DMA builtin, no GWS -- which `grep m0` on the generated assembly shows)
        const unsigned lds0 = (unsigned)(size_t)(const void __attribute__((address_space(3)))*)w_xs;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float* xp = d.x + w_row[t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)] + xo;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds0 + t * 256)), "v"(xp) : "memory");
          if (vec) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds0 + (8 + t) * 256)), "v"(xp + 1) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds0 + (16 + t) * 256)), "v"(xp + 2) : "memory");
          }
        }
      };
      // every load issued before the `younger` most recent ones has landed -- the stage's gathers among them
      auto wait_stage = [&](auto younger_c) {
        constexpr int y = decltype(younger_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (y == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      };
      auto finish_Y = [&](int ct) {
        if (nseg_u == 0 || (ABL & 16)) return;
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), n = lane & 15, g = lane >> 4;
        const unsigned cd_n = b_cdesc[(W.ct0[io] + ct) * 16 + n];
        const bool vec = (__builtin_amdgcn_readfirstlane(cd_n) >> 12) & 1u;
        const int so = (cd_n >> 16) & 15;
        const float yv = (cd_n >> 31) ? sY : 0.f;            // (padding columns: zero)
        float y[8];
        if (!vec) {                                          // x[u] sh0
          float xs[8], s0[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) { xs[t] = w_xs[t * 64 + lane]; s0[t] = w_sh[(t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)) * 12 + so]; }
#pragma unroll
          for (int t = 0; t < 8; ++t) y[t] = xs[t] * s0[t] * yv;
        } else {                                             // xv[u] . sh1
          float xs[8][3], s1[8][3];
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int m = 0; m < 3; ++m) { xs[t][m] = w_xs[(8 * m + t) * 64 + lane]; s1[t][m] = w_sh[(t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)) * 12 + so + m]; }
#pragma unroll
          for (int t = 0; t < 8; ++t) y[t] = (xs[t][0] * s1[t][0] + xs[t][1] * s1[t][1] + xs[t][2] * s1[t][2]) * yv;
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
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define SL __builtin_amdgcn_sched_barrier(0)
      // step A of one (c, k) tile, on its own (the first tile of an irrep, the first tile of a c tile): Z[c, k] of my four segment slots -- independent
      // three-product chains side by side -- as fp32 to LDS (column = segment)
      auto stepA = [&](auto kt_c, auto buf_c) {
        constexpr int kt = decltype(kt_c)::value, boff = decltype(buf_c)::value * CZ_BUF;
        if (ABL & 4) return;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f16x8 hh = __builtin_bit_cast(f16x8, Hh[kt][0]), hl = __builtin_bit_cast(f16x8, Hh[kt][1]);
        f32x4 z[CZ_MAXSEG];
#pragma unroll
        for (int j = 0; j < CZ_MAXSEG; ++j) z[j] = MF(YM(j, 0), hl, zero);
#pragma unroll
        for (int j = 0; j < CZ_MAXSEG; ++j) z[j] = MF(YM(j, 1), hh, z[j]);
#pragma unroll
        for (int j = 0; j < CZ_MAXSEG; ++j) z[j] = MF(YM(j, 0), hh, z[j]);
#pragma unroll
        for (int j = 0; j < CZ_MAXSEG; ++j) *reinterpret_cast<f32x4*>(za[j] + boff) = z[j];
      };
      // x 2^-20 (|Z| <= 32 x 2^15 x 2^15), cut into two fp16 pieces: the B operand of my k-step (a quarter of a column block's eight values per call)
      auto cut = [&](const f32x4 (&zf)[2], u32x4& zh, u32x4& zl, auto r_c) {
        constexpr int r = decltype(r_c)::value;
        unsigned hi, lo;
        cz_split2(zf[r >> 1][2 * (r & 1)] * zs, zf[r >> 1][2 * (r & 1) + 1] * zs, hi, lo);
        zh[r] = hi; zl[r] = lo;
      };
      using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
      using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
      // One tile kt < 9 as ONE instruction stream: step A of tile kt + 1 (12 MFMAs, into the other Z buffer) with the cutting of this tile's first
      // column block in its shadow, then step B of this tile (9 MFMAs per column block) with A's four LDS writes and the cutting of the second block in
      // its shadow.  (Round 5: as separate phases -- A as a loop over segment pairs with a mask read and eight v_and each, B block by block behind
      // scalar branches -- a wave spent 480 + 900 cycles per tile on 30 MFMAs = 480 pipe cycles.)
      auto tile = [&](auto kt_c, auto ncb_c) {
        constexpr int kt = decltype(kt_c)::value, NCBV = decltype(ncb_c)::value;
        constexpr int boffA = ((kt + 1) & 1) * CZ_BUF;
        // (the next tile's fragments are requested behind step A's first CZ_FETCH_AT matrix instructions, not at the top of the tile: same-box A/B, TUNING_r5.md)
        if constexpr (CZ_FETCH_AT == 0) fetchW(std::integral_constant<int, (kt + 1) & 1>{}, min(gq + 1, gq_last));
        const char* zr = zb + (kt & 1) * CZ_BUF + wave * CZ_VSTRIDE + g * 512 + n * 16;
        f32x4 zf0[2], zf1[2];
        zf0[0] = *reinterpret_cast<const f32x4*>(zr); zf0[1] = *reinterpret_cast<const f32x4*>(zr + 256);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f16x8 hh = __builtin_bit_cast(f16x8, Hh[kt + 1][0]), hl = __builtin_bit_cast(f16x8, Hh[kt + 1][1]);
        u32x4 p0h, p0l, p1h, p1l;
        f32x4 z0, z1, z2, z3;
        if constexpr ((ABL & 4) != 0) {
          z0 = z1 = z2 = z3 = zero;
          cut(zf0, p0h, p0l, K0{}); cut(zf0, p0h, p0l, K1{}); cut(zf0, p0h, p0l, K2{}); cut(zf0, p0h, p0l, K3{});
        } else {
        CZ_SETPRIO(3);                       // (falling priority through the tile: of the two waves of a SIMD the one behind wins)
        SL;
        z0 = MF(YM(0, 0), hl, zero); SL;
        z1 = MF(YM(1, 0), hl, zero); SL;
        z2 = MF(YM(2, 0), hl, zero); SL;
        z3 = MF(YM(3, 0), hl, zero); SL;
        if constexpr (CZ_FETCH_AT == 4) { fetchW(std::integral_constant<int, (kt + 1) & 1>{}, min(gq + 1, gq_last)); SL; }
        z0 = MF(YM(0, 1), hh, z0); cut(zf0, p0h, p0l, K0{}); SL;
        z1 = MF(YM(1, 1), hh, z1); SL;
        z2 = MF(YM(2, 1), hh, z2); cut(zf0, p0h, p0l, K1{}); SL;
        z3 = MF(YM(3, 1), hh, z3); SL;
        if constexpr (CZ_FETCH_AT == 8) { fetchW(std::integral_constant<int, (kt + 1) & 1>{}, min(gq + 1, gq_last)); SL; }
        CZ_SETPRIO(2);
        z0 = MF(YM(0, 0), hh, z0); cut(zf0, p0h, p0l, K2{}); SL;
        z1 = MF(YM(1, 0), hh, z1); SL;
        z2 = MF(YM(2, 0), hh, z2); cut(zf0, p0h, p0l, K3{}); SL;
        z3 = MF(YM(3, 0), hh, z3); SL;
        if constexpr (CZ_FETCH_AT == 12) { fetchW(std::integral_constant<int, (kt + 1) & 1>{}, min(gq + 1, gq_last)); SL; }
        }
        if constexpr ((ABL & 4) != 0 && CZ_FETCH_AT != 0) fetchW(std::integral_constant<int, (kt + 1) & 1>{}, min(gq + 1, gq_last));
#define WFR(wt, pc) __builtin_bit_cast(f16x8, Wf[kt & 1][wt][pc])
        if constexpr ((ABL & 8) != 0) {
          if constexpr (!(ABL & 4)) { *reinterpret_cast<f32x4*>(za[0] + boffA) = z0; *reinterpret_cast<f32x4*>(za[1] + boffA) = z1; *reinterpret_cast<f32x4*>(za[2] + boffA) = z2; *reinterpret_cast<f32x4*>(za[3] + boffA) = z3; }
          asm volatile("" :: "v"(p0h), "v"(p0l));
        } else {
          const f16x8 zh = __builtin_bit_cast(f16x8, p0h), zl = __builtin_bit_cast(f16x8, p0l);
          if constexpr (NCBV > 1) { zf1[0] = *reinterpret_cast<const f32x4*>(zr + CZ_CB); zf1[1] = *reinterpret_cast<const f32x4*>(zr + CZ_CB + 256); }
          acc[0][0] = MF(WFR(0, 0), zl, acc[0][0]); *reinterpret_cast<f32x4*>(za[0] + boffA) = z0; SL;
          acc[1][0] = MF(WFR(1, 0), zl, acc[1][0]); *reinterpret_cast<f32x4*>(za[1] + boffA) = z1; SL;
          acc[2][0] = MF(WFR(2, 0), zl, acc[2][0]); *reinterpret_cast<f32x4*>(za[2] + boffA) = z2; SL;
          acc[0][0] = MF(WFR(0, 1), zh, acc[0][0]); *reinterpret_cast<f32x4*>(za[3] + boffA) = z3; SL;
          CZ_SETPRIO(1);
          acc[1][0] = MF(WFR(1, 1), zh, acc[1][0]); if constexpr (NCBV > 1) cut(zf1, p1h, p1l, K0{}); SL;
          acc[2][0] = MF(WFR(2, 1), zh, acc[2][0]); if constexpr (NCBV > 1) cut(zf1, p1h, p1l, K1{}); SL;
          acc[0][0] = MF(WFR(0, 0), zh, acc[0][0]); if constexpr (NCBV > 1) cut(zf1, p1h, p1l, K2{}); SL;
          acc[1][0] = MF(WFR(1, 0), zh, acc[1][0]); if constexpr (NCBV > 1) cut(zf1, p1h, p1l, K3{}); SL;
          acc[2][0] = MF(WFR(2, 0), zh, acc[2][0]); SL;
        }
        if constexpr (NCBV > 1 && !(ABL & 8)) {
          const f16x8 zh = __builtin_bit_cast(f16x8, p1h), zl = __builtin_bit_cast(f16x8, p1l);
          acc[0][1] = MF(WFR(0, 0), zl, acc[0][1]); SL;
          acc[1][1] = MF(WFR(1, 0), zl, acc[1][1]); SL;
          acc[2][1] = MF(WFR(2, 0), zl, acc[2][1]); SL;
          CZ_SETPRIO(0);
          acc[0][1] = MF(WFR(0, 1), zh, acc[0][1]); SL;
          acc[1][1] = MF(WFR(1, 1), zh, acc[1][1]); SL;
          acc[2][1] = MF(WFR(2, 1), zh, acc[2][1]); SL;
          acc[0][1] = MF(WFR(0, 0), zh, acc[0][1]); SL;
          acc[1][1] = MF(WFR(1, 0), zh, acc[1][1]); SL;
          acc[2][1] = MF(WFR(2, 0), zh, acc[2][1]); SL;
        }
      };
      // the k tile 9 (the bias row: k-step 0 only, wave 0's) and, under it, the next c tile's Y and its first step A
      auto tile9 = [&](int ct) {
        constexpr int kt = KT;
        // Order: wave 0's step B, the next c tile's Y and first step A, and only then the W2' fetch of the next tile.  Both parts reload a few spilled
        // loop invariants, and a reload waits for vmcnt(0): behind the fetch that is a whole L2 round trip (7 800 cycles for this tile when the
        // fetch came first).
        wait_stage(K0{});
        if (wave == 0 && !(ABL & 8)) {
          const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), n = lane & 15, g = lane >> 4;
          const char* zr = zb + (kt & 1) * CZ_BUF + g * 512 + n * 16;
#pragma unroll
          for (int cb = 0; cb < CZ_NCB; ++cb)
            if (cb < ncb) {
              f32x4 zf[2];
              zf[0] = *reinterpret_cast<const f32x4*>(zr + cb * CZ_CB); zf[1] = *reinterpret_cast<const f32x4*>(zr + cb * CZ_CB + 256);
              u32x4 ph, pl;
              cut(zf, ph, pl, K0{}); cut(zf, ph, pl, K1{}); cut(zf, ph, pl, K2{}); cut(zf, ph, pl, K3{});
              const f16x8 zh = __builtin_bit_cast(f16x8, ph), zl = __builtin_bit_cast(f16x8, pl);
#pragma unroll
              for (int wt = 0; wt < 3; ++wt) {
                acc[wt][cb] = MF(WFR(wt, 0), zl, acc[wt][cb]);
                acc[wt][cb] = MF(WFR(wt, 1), zh, acc[wt][cb]);
                acc[wt][cb] = MF(WFR(wt, 0), zh, acc[wt][cb]);
              }
            }
        }
        if (ct + 1 < W.nct[io]) { finish_Y(ct + 1); stepA(K0{}, K0{}); }
        __builtin_amdgcn_sched_barrier(0);
        fetchW(K0{}, min(gq + 1, gq_last));
      };
#undef WFR
      // While step B reads tile i from buffer i & 1, step A of tile i + 1 is written into the other buffer; one barrier per tile.
      prefetch_x(0);
      wait_stage(K0{});
      fetchW(K0{}, gq);
      finish_Y(0);
      stepA(K0{}, K0{});
      __syncthreads();
      // (the column-block count is decided once per unit, OUTSIDE the tile loops: as a branch inside a tile hipcc hoists the instructions the two
      // forms share -- the Z reads and the first half of the cutting, with their s_waitcnt -- in front of step A's matrix instructions)
      auto run_tiles = [&](auto ncb_c) {
        for (int ct = 0; ct < W.nct[io]; ++ct) {
          cz_static_for<0, CZ_NKT>([&](auto kt_c) {
            constexpr int kt = decltype(kt_c)::value;
            stamp(10);
            if constexpr (kt < KT) {
              tile(kt_c, ncb_c);
              // (behind the tile's matrix instructions, where the wave would wait at the barrier; used at kt = 9)
              if constexpr (kt == CZ_PF_KT) if (ct + 1 < W.nct[io]) prefetch_x(ct + 1);
            } else
              tile9(ct);
            stamp(12);
            ++gq;
            if (!(ABL & 2)) __syncthreads();                 // the next tile's Z is complete; this tile's buffer may be written again
            stamp(13);
          });
        }
      };
      if (ncb > 1) run_tiles(K2{}); else run_tiles(K1{});
#undef YM
#undef MF
#undef SL
      // ---- the output irrep is complete: add the eight waves' partial sums, take the factors off, store into the segments' first rows
      float* red = reinterpret_cast<float*>(zb) + wave * (12 * CZ_NCB * 64);
#pragma unroll
      for (int wt = 0; wt < 3; ++wt)
#pragma unroll
        for (int cb = 0; cb < CZ_NCB; ++cb)
#pragma unroll
          for (int q = 0; q < 4; ++q) red[((wt * CZ_NCB + cb) * 4 + q) * 64 + lane] = acc[wt][cb][q];
      __syncthreads();
      for (int idx = tid; idx < 16 * ncb * 48; idx += 64 * NW) {
        const int col = idx / 48, w = idx - 48 * col;
        const int e = b_col_edge[col];
        if (e >= 0) {
          const int r = w & 15;
          const float* rp = reinterpret_cast<const float*>(zb) + (((w >> 4) * CZ_NCB + (col >> 4)) * 4 + (r & 3)) * 64 + 16 * (r >> 2) + (col & 15);
          float s = 0.f;
#pragma unroll
          for (int v = 0; v < NW; ++v) s += rp[v * (12 * CZ_NCB * 64)];
          d.msg[(size_t)e * d.D_out + W.out_off[io] + w] = s * b_col_inv[col] * W.rowinv[io * 48 + w];
        }
      }
      __syncthreads();
    }
    if (a.executed && lane == 0) atomicAdd(a.executed, 16384.0 * (double)n_mfma);   // 16 x 16 x 32 x 2 flops per instruction
  }
}

// (profiling only, its own launch -- k_convz itself is not touched: a handful of extra live values moved its spills and cost 14 % of its speed)
// The flops among those k_convz issues that are not padding, [1] of the counters, and the message bytes of the form that runs, [2]: per chunk the hidden
// layer of its edges; step A the edges x the (path, u) pairs that exist x 145 (the k tile 9: one column); step B one column per segment x (c, k) values
// x 48; three partial products each.  One thread per chunk.
__global__ void k_convz_useful(ConvZArgs a) {
  for (int c = 0; c < a.n_conv; ++c) {
    const ConvZDesc& d = a.c[c];
    const ConvZ& W = d.w;
    const int nch = min(*d.n_chunks, d.max_chunks), E = min(*d.n_edges, d.max_edges);
    double useful = 0.0, bytes = 0.0;
    for (int ch = blockIdx.x * blockDim.x + threadIdx.x; ch < nch; ch += gridDim.x * blockDim.x) {
      const int es = d.chunk_es[ch], len = min(min(d.chunk_gl[ch] & 63, 32), max(E - es, 0));
      int nseg = 0;
      for (int i = 0; i < len; ++i) nseg += (i == 0 || d.tgt[es + i] != d.tgt[es + i - 1]);
      nseg = min(nseg, CZ_MAXSEG);
      double u = 144.0 * 144.0 * len;
      for (int io = 0; io < W.n_io; ++io) u += (double)W.nc_valid[io] * 145.0 * ((double)len + 48.0 * nseg);
      useful += 6.0 * u;
      bytes += 4.0 * 48.0 * W.n_io * nseg;
    }
    if (useful > 0.0) { atomicAdd(a.executed + 1, useful); atomicAdd(a.executed + 2, bytes); }
  }
}

size_t convz_lds_bytes() { return CZ_R0BYTES + 8 * CZ_WAVE_FLOATS * sizeof(float) + (8 + 2 * 16 * CZ_NCB + CZ_MAXCT * 16) * sizeof(int); }

void launch_convz(const ConvZArgs& a0, hipStream_t st) {
  constexpr int NW = 8;
  ConvZArgs a = a0;
  static float* dbg_dev = nullptr;
  static const char* dbg_file = getenv("DBFR_CONVZ_DEBUG");
  if (dbg_file && !dbg_dev) {
    if (hipMalloc(&dbg_dev, 8 * 1024 * 8) != hipSuccess || hipMemset(dbg_dev, 0, 8 * 1024 * 8) != hipSuccess) dbg_dev = nullptr;
    else atexit([] {
      std::vector<float> h(8 * 1024 * 2);
      if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h.data(), dbg_dev, h.size() * 4, hipMemcpyDeviceToHost) == hipSuccess)
        if (FILE* f = fopen(getenv("DBFR_CONVZ_DEBUG"), "wb")) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
    });
  }
  a.dbg = dbg_dev;
  const size_t lds = convz_lds_bytes();
#define V(x) { if (dbfr_launch_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_convz<NW, x>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "k_convz: hipFuncSetAttribute(MaxDynamicSharedMemorySize)")) return; \
               hipLaunchKernelGGL((k_convz<NW, x>), dim3(dbfr_current_cu_count()), dim3(64 * NW), lds, st, a); return; }
#ifdef DBFR_DEV_VARIANTS
  static int abl = getenv("DBFR_CONVZ_ABL") ? atoi(getenv("DBFR_CONVZ_ABL")) : 0;
  if (abl == 1) V(1) if (abl == 2) V(2) if (abl == 4) V(4) if (abl == 8) V(8) if (abl == 16) V(16) if (abl == 32) V(32) if (abl == 12) V(12) if (abl == 5) V(5) if (abl == 64) V(64) if (abl == 128) V(128)
#endif
  if (a.executed) hipLaunchKernelGGL(k_convz_useful, dim3(64), dim3(256), 0, st, a);
  V(0)
#undef V
}
