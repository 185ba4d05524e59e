// Reduce-first form of the SCALAR-OUTPUT tensor-product paths of a K = 144 conv (DBFR_GEMM_REDUCE_FIRST) -- round 6, form D: sixteen chunks per W2' tile.
// Algorithm, scaling and message interface as in round 5's kernel (diffbindfr_amd/csrc/convz.hip, docs/kernels/conv_reduce_first.md):
//     Z[t, c, k] = sum_{e -> t} y[e, c] h'[e, k]     (step A, per chunk of <= 32 edges / <= 4 targets, segment = target inside a chunk)
//     out_sum[t, o] = sum_{c,k} Z[t, c, k] W2'[c, k, o]   (step B, columns = segments)
// What bounds the tile loop is the stream of W2' fragments out of L2: 48 KB per (c, k) tile and workgroup at ~26 B per cycle, a per-CU limit
// (profiles/TUNING_r6.md).  The lever is more chunks per fragment tile; a chunk's state is its hidden layer H (80 registers per lane).  Here:
//   * TWELVE waves in two roles, 168 registers each: eight CHUNK waves with TWO chunks each (16 chunks = 64 columns per W2' tile, twice round 5's) and four
//     COLUMN waves (two k-steps of step B each, two fragment tiles in registers: this tile's and the next one's in flight);
//   * the k range in TWO PASSES (k tiles 0..4, then 5..8 and the bias tile): half of H per chunk at a time (40 registers, 80 for the wave's two chunks); the other
//     half is computed at the start of the second pass (W1h half by half into LDS, the inputs loaded again) and the second pass's partial sums are ADDED to
//     those the first one stored (same workgroup, fixed order: no atomics, batch-independent);
//   * Y per chunk UNMASKED in registers (8), the segment masks applied per step A from LDS (the masked copies, 32 registers per chunk, do not fit twice);
//   * a tile in TWO PHASES: the chunks form two groups of eight (first / second chunk of every wave) with one SINGLE-buffered Z region each; while the column
//     waves work through group 0 of tile t the chunk waves write group 1 of tile t, then the column waves take group 1 while group 0 of tile t + 1 is
//     written -- a region is never read and written in the same phase, and 64 columns of Z take the LDS 32 took double-buffered.
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

#define CZ_NWA 8                    // chunk waves
#define CZ_NC 2                     // chunks per chunk wave = groups
#define CZ_UNIT (CZ_NWA * CZ_NC)    // chunks per unit
#define CZ_NWB 4                    // column waves
#define CZ_NW (CZ_NWA + CZ_NWB)
#define CZ_KS (8 / CZ_NWB)          // k-steps of 32 (of the 256 (c, k) values of a tile) per column wave
#define CZ_KTP 5                    // k tiles per pass

// Z in LDS, already cut into fp16 pieces by the chunk wave that computed it.  Per group and column block two planes (hi, lo), each
// [k-step v: 1040 B][lane group g'': 256 B][column: 16 B = (k_local v: c 4g''..4g''+3 | k_local v + 8: the same c)].  Step A's lane (k_local n, group g) writes
// its four values c_local = 4 g + q as one 8-byte store per plane to (v = n & 7, g'' = g, half n >> 3); step B's lane (column, g'') reads the 16 bytes of its
// k-step from each plane: they ARE the B operand.  The 1040-byte stride (4 dwords mod 32 banks) makes the stores conflict-free, the reads are contiguous.
#define CZ_VSTRIDE 1040
#define CZ_PLANE (8 * CZ_VSTRIDE)
#define CZ_CB (2 * CZ_PLANE)        // one column block: hi plane, lo plane
#define CZ_GCB 2                    // column blocks of 16 per group: the <= 8 x CZ_MAXSEG segments of a group's chunks take consecutive columns
#define CZ_NCB (CZ_NC * CZ_GCB)
#define CZ_GBYTES (CZ_GCB * CZ_CB)  // a group's region
#define CZ_ZBYTES (CZ_NC * CZ_GBYTES)
#define CZ_CHUNK_FLOATS (32 * 12 + 32 + 32 + 32 + 32 + 32 + CZ_MAXSEG * 16)   // per chunk: harmonics [32][12] | sa | ua | gather row offsets | segment ids | first slot of segment j | masks [CZ_MAXSEG segments][4 lane groups][4 dwords]
#define CZ_BLOCK_INTS (CZ_UNIT + 2 * 16 * CZ_NCB + CZ_MAXCT * 16)                 // b_nseg | b_col_edge | b_col_inv | b_cdesc
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
// too), P2 (hidden layer done: the Z regions are free; the first pass: segment counts written), then per output irrep B3 (group 0 of the first tile written),
// two per (c, k) tile (B4a: group 1 of the tile written / group 0 read; B4b: group 0 of the next tile written / group 1 read), B5 (partial sums written), B6 (stored).
template <int ABL = 0>   // ABL (developer, wrong results): 4 no step A, 8 no step B; 128 (right results): s_memtime timeline of one unit of workgroup 0
__global__ __launch_bounds__(64 * CZ_NW) void k_convz(ConvZArgs a) {
  constexpr int KT = 9, NWA = CZ_NWA, NC = CZ_NC, UNIT = CZ_UNIT, NWB = CZ_NWB, KS = CZ_KS, NCB = CZ_NCB, GCB = CZ_GCB, KTP = CZ_KTP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave: a scalar register)
  const int n = lane & 15, g = lane >> 4;
  char* zb = reinterpret_cast<char*>(lds);
  int* b_nseg = reinterpret_cast<int*>(lds + CZ_ZBYTES / 4 + UNIT * CZ_CHUNK_FLOATS);   // [UNIT] chunk (group, wave) at group * NWA + wave
  int* b_col_edge = b_nseg + UNIT;                          // [16 NCB] message row of the column's segment, -1: column unused
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
  // ---- unit list: UNIT chunks per unit, conv after conv
  int nch[4] = {0, 0, 0, 0}, nu[4] = {0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < a.n_conv) {
      const ConvZDesc& d = a.c[c];
      nch[c] = min(*d.n_chunks, d.max_chunks);
      nu[c] = (nch[c] + UNIT - 1) / UNIT;
    }
  const int N = nu[0] + nu[1] + nu[2] + nu[3];
  // an output irrep of a pass is complete: add the column waves' partial sums (B5 behind them), take the factors off, store into the segments' first rows (the
  // second pass: add to what the first one stored) -- all twelve waves
  auto store_irrep = [&](const ConvZDesc& d, const ConvZ& W, int io, int pass) {
    for (int idx = tid; idx < 16 * NCB * 48; idx += 64 * CZ_NW) {
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
    // =============================================================== chunk waves: wave w owns chunk w of group 0 and chunk w of group 1
    float* cl0 = lds + CZ_ZBYTES / 4 + wave * (NC * CZ_CHUNK_FLOATS);
#define C_SH(gr) (cl0 + (gr) * CZ_CHUNK_FLOATS)                                  /* [32][12] harmonics */
#define C_SA(gr) (C_SH(gr) + 32 * 12)                                            /* [32] the edge's factor on its radial-MLP inputs */
#define C_UA(gr) (C_SA(gr) + 32)                                                 /* [32] ... and its inverse */
#define C_ROW(gr) (reinterpret_cast<int*>(C_UA(gr) + 32))                        /* [32] gth[e] * ldx */
#define C_SEG(gr) (C_ROW(gr) + 32)                                               /* [32] segment of the slot, -1: no edge */
#define C_FIRST(gr) (C_SEG(gr) + 32)                                             /* [32] first slot of segment j */
#define C_MASK(gr) (reinterpret_cast<unsigned*>(C_FIRST(gr) + 32))               /* [CZ_MAXSEG][4][4] segment j's mask on the A-operand registers of lane group g */
    for (int unit = blockIdx.x; unit < N; unit += gridDim.x) {
      int c = 0, ul = unit;
      if (ul >= nu[0]) { ul -= nu[0]; c = 1; if (ul >= nu[1]) { ul -= nu[1]; c = 2; if (ul >= nu[2]) { ul -= nu[2]; c = 3; } } }
      const ConvZDesc& d = a.c[c];
      const ConvZ& W = d.w;
      const int E = min(*d.n_edges, d.max_edges);
      tr_on = (ABL & 128) && blockIdx.x == 0 && unit == 2 * (int)gridDim.x && (a.dbg_sel == 0 || a.dbg_sel == a.c[0].w.ct0[a.c[0].w.n_io - 1] + a.c[0].w.nct[a.c[0].w.n_io - 1]);
      if (tr_on) tr_n = 0;
      stamp(1);
      // ---- my chunks: edges [es, es + len) each (k_chunk_fill: <= 32 edges and <= CZ_MAXSEG targets of one graph); group gr's chunk of this wave is chunk
      // ul UNIT + gr NWA + wave of the conv
      int es[NC], len[NC], nsg[NC], ey[NC], e_et[NC][2], i1[NC][2], i2[NC][2];
      cz_static_for<0, NC>([&](auto gr_c) {
        constexpr int gr = decltype(gr_c)::value;
        const int ch = ul * UNIT + gr * NWA + wave;
        int es_ = 0, len_ = 0, gidx = 0;
        if (ch < nch[c]) {
          es_ = d.chunk_es[ch];
          const int gl = d.chunk_gl[ch];
          gidx = gl >> 6;
          len_ = min(min(gl & 63, 32), max(E - es_, 0));
        }
        es[gr] = es_; len[gr] = len_;
        // slots: lanes 0..31 own slot L (clamped to the chunk's last edge beyond its length; a chunk without edges reads nothing)
        const int sl = lane & 31;
        const bool have = len_ > 0;
        const int e_sl = have ? es_ + min(sl, len_ - 1) : 0;
        // (my edge of the two edge tiles of the hidden layer, and its radial-MLP rows: the indices are requested here, with the slots' loads)
        e_et[gr][0] = have ? es_ + min(n, len_ - 1) : 0; e_et[gr][1] = have ? es_ + min(16 + n, len_ - 1) : 0;
        int tgt_l = 0, gth_l = 0;
        i1[gr][0] = i1[gr][1] = i2[gr][0] = i2[gr][1] = 0;
        float shv[9];
        float xmx = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) shv[k] = 0.f;
        if (have) {
          tgt_l = d.tgt[e_sl]; gth_l = d.gth[e_sl];
          i1[gr][0] = d.idx1[e_et[gr][0]]; i1[gr][1] = d.idx1[e_et[gr][1]]; i2[gr][0] = d.idx2[e_et[gr][0]]; i2[gr][1] = d.idx2[e_et[gr][1]];
          if (lane < 32) {
            const float* sp = d.sh + (size_t)e_sl * SH_LD;
#pragma unroll
            for (int k = 0; k < 9; ++k) shv[k] = sp[k];
          }
          if (d.xmax) xmx = d.xmax[gidx];                     // per graph, computed once per layer (k_row_absmax)
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
        const bool is_first = lane < 32 && sl < len_ && (sl == 0 || tgt_l != tgt_prev);
        const unsigned firsts = (unsigned)__ballot(is_first);
        const int nseg = min(__popc(firsts), CZ_MAXSEG);       // (the chunk table holds no chunk with more)
        const int seg_l = (sl < len_) ? __popc(firsts & (0xffffffffu >> (31 - sl))) - 1 : -1;
        float smx = 0.f;
        if (lane < 32) {
          C_ROW(gr)[sl] = gth_l * d.ldx;
          C_SEG(gr)[sl] = seg_l;
          if (is_first) C_FIRST(gr)[seg_l] = sl;
#pragma unroll
          for (int k = 0; k < 9; ++k) { C_SH(gr)[sl * 12 + k] = shv[k]; smx = fmaxf(smx, fabsf(shv[k])); }
          C_SH(gr)[sl * 12 + 9] = C_SH(gr)[sl * 12 + 10] = C_SH(gr)[sl * 12 + 11] = 0.f;
        }
        if (lane == 0) b_nseg[gr * NWA + wave] = nseg;
        // (the scalar columns of the message rows that are not the first of their segment are NOT written: the reductions read those columns of flagged
        // rows only -- EdgeSet::seg_first, conv.hip row_sum)
        xmx = cz_wave_max(xmx); smx = cz_wave_max(smx);
        int ey_ = 0;                                           // y is multiplied by 2^ey: |y| <= 3 |x| |sh| < 2^(ex + es + 2) -> below 2^15
        if (xmx > 0.f && smx > 0.f) ey_ = max(-100, min(100, 13 - __builtin_amdgcn_frexp_expf(xmx) - __builtin_amdgcn_frexp_expf(smx)));
        ey[gr] = __builtin_amdgcn_readfirstlane(ey_);
        nsg[gr] = __builtin_amdgcn_readfirstlane(nseg);
        __builtin_amdgcn_wave_barrier();
        // masks of my segments on the registers of the A operand (halves of a dword = two consecutive slots), for every lane group
        {                                                      // (segments the chunk does not have: all-zero masks)
          const int j = lane >> 4, gg = (lane >> 2) & 3, r = lane & 3, t0 = 2 * r;
          const int s0 = t0 < 4 ? 4 * gg + t0 : 16 + 4 * gg + (t0 - 4);
          C_MASK(gr)[lane] = (C_SEG(gr)[s0] == j ? 0x0000ffffu : 0u) | (C_SEG(gr)[s0 + 1] == j ? 0xffff0000u : 0u);
        }
      });
      stamp(2);
      const float zs = __builtin_amdgcn_ldexpf(1.f, CZ_ZSCALE);
      int cbase[NC] = {0, 0};
      char* za[NC][CZ_MAXSEG];
      for (int pass = 0; pass < 2; ++pass) {
        // ---- hidden layer of this pass's unit tiles m = 5 pass + jj (jj < 5; the second pass has four, its fifth k tile carries the bias), transposed:
        // D[edge, unit] = sum_f a[edge, f] W1[unit, f]; A = the edge's inputs (cut per edge), B = W1h tiles from LDS; chunk after chunk, edge tile after edge tile
        u32x4 Hh[NC][KTP][2];                                  // H pieces [chunk][k tile of the pass][hi, lo]: lane (unit n, group g), eight edges {4g..4g+3, 16+4g..16+4g+3}
        int ephi[NC];                                          // log2 of the factor on h (phi, also what the constant 1 of the bias becomes)
        // raw radial-MLP inputs of my edge of an edge tile: [emb | tab1 row | tab2 row], 16 floats of each per lane group
        // (one copy of the layer's code for both chunks: a loop over the chunk with its per-chunk values picked by the loop variable, the result copied into the
        // chunk's registers behind a uniform branch -- unrolled twice it is 9 KB more code than the instruction cache cares for)
        auto load_inputs = [&](int gr, int et, f32x4 (&Ba)[KT]) {
          const bool have = (gr ? len[1] : len[0]) > 0;
          const int e_ = gr ? (et ? e_et[1][1] : e_et[1][0]) : (et ? e_et[0][1] : e_et[0][0]);
          const int a_ = gr ? (et ? i1[1][1] : i1[1][0]) : (et ? i1[0][1] : i1[0][0]);
          const int b_ = gr ? (et ? i2[1][1] : i2[1][0]) : (et ? i2[0][1] : i2[0][0]);
          const float* r0 = d.emb + (size_t)e_ * NS;
          const float* r1 = d.tab1 + (size_t)a_ * d.ld1;
          const float* r2 = d.tab2 + (size_t)b_ * d.ld2;
#pragma unroll
          for (int s4 = 0; s4 < KT; ++s4) {
            const float* src = s4 < 3 ? r0 : s4 < 6 ? r1 : r2;
            Ba[s4] = have ? *reinterpret_cast<const f32x4*>(src + 16 * (s4 % 3) + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        };
        f32x4 Ba0[KT];
        load_inputs(0, 0, Ba0);
        __syncthreads();   // P1: this pass's half of W1h is in LDS (tile jj at jj x 9280), and the c-tile descriptors (column waves)
        stamp(4);
#pragma unroll 1
        for (int gr = 0; gr < NC; ++gr) {
          float* c_sa = cl0 + gr * CZ_CHUNK_FLOATS + 32 * 12;
          float* c_ua = c_sa + 32;
          float Hf[KTP][2][4];
          float hmx = 0.f;
          cz_static_for<0, 2>([&](auto et_c) {
            constexpr int et = decltype(et_c)::value;
            f32x4 Ba[KT];
            if (et == 0 && gr == 0) {
#pragma unroll
              for (int s4 = 0; s4 < KT; ++s4) Ba[s4] = Ba0[s4];
            } else load_inputs(gr, et, Ba);
            float amx = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < KT; ++s4)
#pragma unroll
              for (int q = 0; q < 4; ++q) amx = fmaxf(amx, fabsf(Ba[s4][q]));
            amx = fmaxf(amx, __shfl_xor(amx, 16));
            amx = fmaxf(amx, __shfl_xor(amx, 32));
            const int ja = max(-14, __builtin_amdgcn_frexp_expf(amx));
            const float sa = __builtin_amdgcn_ldexpf(1.f, 15 - ja);
            if (g == 0) { c_sa[16 * et + n] = sa; c_ua[16 * et + n] = __builtin_amdgcn_ldexpf(1.f, ja - 15); }
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
            const f32x4 sar = *reinterpret_cast<const f32x4*>(c_sa + 16 * et + 4 * g), uar = *reinterpret_cast<const f32x4*>(c_ua + 16 * et + 4 * g);
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
          int eh = max(hmx > 0.f ? __builtin_amdgcn_frexp_expf(hmx) : W.k1, W.k1);
          eh = __builtin_amdgcn_readfirstlane(min(eh, W.k1 + 100));
          const int ephi_ = 15 - eh + W.k1;
          const float sH = __builtin_amdgcn_ldexpf(1.f, 15 - eh);
          u32x4 Ht[KTP][2];
#pragma unroll
          for (int jj = 0; jj < KTP; ++jj) {
            unsigned h0, l0, h1, l1, h2, l2, h3, l3;
            cz_split2(Hf[jj][0][0] * sH, Hf[jj][0][1] * sH, h0, l0);
            cz_split2(Hf[jj][0][2] * sH, Hf[jj][0][3] * sH, h1, l1);
            cz_split2(Hf[jj][1][0] * sH, Hf[jj][1][1] * sH, h2, l2);
            cz_split2(Hf[jj][1][2] * sH, Hf[jj][1][3] * sH, h3, l3);
            Ht[jj][0] = (u32x4){h0, h1, h2, h3};
            Ht[jj][1] = (u32x4){l0, l1, l2, l3};
          }
          if (pass) {   // the k tile 9 (the second pass's fifth): the constant 1 (x phi) of the bias in column 0, for the slots that hold an edge
            const _Float16 ph = (_Float16)__builtin_amdgcn_ldexpf(1.f, ephi_);
            const unsigned short pb = __builtin_bit_cast(unsigned short, ph);
            const int len_ = gr ? len[1] : len[0];
            u32x4 hb = {0u, 0u, 0u, 0u};
            if (n == 0) {
#pragma unroll
              for (int t = 0; t < 8; ++t) {
                const int slot = t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4);
                if (slot < len_) hb[t >> 1] |= (unsigned)pb << (16 * (t & 1));
              }
            }
            Ht[KTP - 1][0] = hb;
            Ht[KTP - 1][1] = (u32x4){0u, 0u, 0u, 0u};
          }
          if (gr == 0) {
            ephi[0] = ephi_;
#pragma unroll
            for (int jj = 0; jj < KTP; ++jj) { Hh[0][jj][0] = Ht[jj][0]; Hh[0][jj][1] = Ht[jj][1]; }
          } else {
            ephi[1] = ephi_;
#pragma unroll
            for (int jj = 0; jj < KTP; ++jj) { Hh[1][jj][0] = Ht[jj][0]; Hh[1][jj][1] = Ht[jj][1]; }
          }
        }
        stamp(3);
        __syncthreads();   // P2: the Z regions are free again; the segment counts of the chunks and the cleared column table (first pass)
        if (pass == 0) {
          // ---- columns: the segments of a group's chunks side by side (<= 8 x CZ_MAXSEG = 32 = two column blocks per group); group gr's columns start at 32 gr
#pragma unroll
          for (int gr = 0; gr < NC; ++gr) {
            int cb_ = 16 * GCB * gr;
#pragma unroll
            for (int v = 0; v < NWA; ++v) if (v < wave) cb_ += b_nseg[gr * NWA + v];
            cbase[gr] = __builtin_amdgcn_readfirstlane(cb_);
            if (lane < nsg[gr]) b_col_edge[cbase[gr] + lane] = es[gr] + C_FIRST(gr)[lane];
            // LDS addresses of my segments' Z columns, hi plane (lane part + column).  A slot without a segment computes zeros and writes them into the
            // group's last column: unused whenever such a slot exists (the group then has fewer than 32 segments), and never stored
            char* zw0 = zb + (n & 7) * CZ_VSTRIDE + g * 256 + (n >> 3) * 8;
#pragma unroll
            for (int j = 0; j < CZ_MAXSEG; ++j) {
              const int col = cbase[gr] + j;
              za[gr][j] = j < nsg[gr] ? zw0 + (col >> 4) * CZ_CB + (col & 15) * 16 : zw0 + (GCB * gr + GCB - 1) * CZ_CB + 15 * 16;
            }
          }
        }
#pragma unroll
        for (int gr = 0; gr < NC; ++gr)
          if (lane < nsg[gr]) b_col_inv[cbase[gr] + lane] = __builtin_amdgcn_ldexpf(1.f, -CZ_ZSCALE - ey[gr] - ephi[gr]);
        // Y of a c tile per chunk, UNMASKED: lane (c = n, group g), my eight slots, as fp16 pieces; the segment masks are applied in step A.  The x values are
        // gathered straight into registers where they are needed (the chunk waves are ahead of the column waves there).
        u32x4 Yh[NC], Yl[NC];
#pragma unroll
        for (int gr = 0; gr < NC; ++gr) { Yh[gr] = (u32x4){0u, 0u, 0u, 0u}; Yl[gr] = (u32x4){0u, 0u, 0u, 0u}; }
        auto make_Y = [&](auto gr_c, int ctg) {                // ctg: c tile of the conv (over both irreps)
          constexpr int gr = decltype(gr_c)::value;
          if (nsg[gr] == 0) return;
          const unsigned cd_n = b_cdesc[ctg * 16 + n];
          const bool vec = (__builtin_amdgcn_readfirstlane(cd_n) >> 12) & 1u;
          const int xo = cd_n & 0xfff, so = (cd_n >> 16) & 15;
          const float yv = (cd_n >> 31) ? __builtin_amdgcn_ldexpf(1.f, ey[gr]) : 0.f;   // (padding columns: zero)
          float y[8];
          if (!vec) {                                          // x[u] sh0
            float xs[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) xs[t] = d.x[C_ROW(gr)[t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)] + xo];
#pragma unroll
            for (int t = 0; t < 8; ++t) y[t] = xs[t] * C_SH(gr)[(t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)) * 12 + so] * yv;
          } else {                                             // xv[u] . sh1
            float xs[8][3];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const float* xp = d.x + C_ROW(gr)[t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)] + xo;
              xs[t][0] = xp[0]; xs[t][1] = xp[1]; xs[t][2] = xp[2];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const float* sp = C_SH(gr) + (t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)) * 12 + so;
              y[t] = (xs[t][0] * sp[0] + xs[t][1] * sp[1] + xs[t][2] * sp[2]) * yv;
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { unsigned hi, lo; cz_split2(y[2 * r], y[2 * r + 1], hi, lo); Yh[gr][r] = hi; Yl[gr][r] = lo; }
        };
        // step A of one (c, k) tile for the chunk of group gr: Z[c, k] of its four segment slots -- the mask of a segment on Y, a three-product chain, two segments
        // side by side -- x 2^-20 (|Z| <= 32 x 2^15 x 2^15), cut into two fp16 pieces and stored (column = segment): the B operand of step B as the column waves read it
        auto stepA = [&](auto gr_c, auto j_c) {
          constexpr int gr = decltype(gr_c)::value, jj = decltype(j_c)::value;
          if (ABL & 4) return;
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          const f16x8 hh = __builtin_bit_cast(f16x8, Hh[gr][jj][0]), hl = __builtin_bit_cast(f16x8, Hh[gr][jj][1]);
#pragma unroll
          for (int jp = 0; jp < CZ_MAXSEG; jp += 2) {
            const u32x4 m0_ = *reinterpret_cast<const u32x4*>(C_MASK(gr) + (jp * 4 + g) * 4), m1_ = *reinterpret_cast<const u32x4*>(C_MASK(gr) + ((jp + 1) * 4 + g) * 4);
            const f16x8 a0h = __builtin_bit_cast(f16x8, Yh[gr] & m0_), a0l = __builtin_bit_cast(f16x8, Yl[gr] & m0_);
            const f16x8 a1h = __builtin_bit_cast(f16x8, Yh[gr] & m1_), a1l = __builtin_bit_cast(f16x8, Yl[gr] & m1_);
            f32x4 z0 = MF(a0h, hl, zero), z1 = MF(a1h, hl, zero);
            z0 = MF(a0l, hh, z0); z1 = MF(a1l, hh, z1);
            z0 = MF(a0h, hh, z0); z1 = MF(a1h, hh, z1);
            unsigned h0, l0, h1, l1;
            cz_split2(z0[0] * zs, z0[1] * zs, h0, l0);
            cz_split2(z0[2] * zs, z0[3] * zs, h1, l1);
            *reinterpret_cast<u32x2*>(za[gr][jp]) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(za[gr][jp] + CZ_PLANE) = (u32x2){l0, l1};
            cz_split2(z1[0] * zs, z1[1] * zs, h0, l0);
            cz_split2(z1[2] * zs, z1[3] * zs, h1, l1);
            *reinterpret_cast<u32x2*>(za[gr][jp + 1]) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(za[gr][jp + 1] + CZ_PLANE) = (u32x2){l0, l1};
          }
        };
        using G0 = std::integral_constant<int, 0>; using G1 = std::integral_constant<int, 1>;
        for (int io = 0; io < W.n_io; ++io) {
          const int ct0 = W.ct0[io];
          make_Y(G0{}, ct0); make_Y(G1{}, ct0);
          stepA(G0{}, std::integral_constant<int, 0>{});
          __syncthreads();   // B3
          for (int ct = 0; ct < W.nct[io]; ++ct) {
            cz_static_for<0, KTP>([&](auto j_c) {
              constexpr int jj = decltype(j_c)::value;
              stamp(10);
              stepA(G1{}, j_c);                                // group 1 of this tile, while the column waves read group 0
              stamp(11);
              __syncthreads();   // B4a
              if constexpr (jj + 1 < KTP) stepA(G0{}, std::integral_constant<int, jj + 1>{});   // group 0 of the next tile, while they read group 1
              else if (ct + 1 < W.nct[io]) { make_Y(G0{}, ct0 + ct + 1); make_Y(G1{}, ct0 + ct + 1); stepA(G0{}, std::integral_constant<int, 0>{}); }
              stamp(12);
              __syncthreads();   // B4b
              stamp(13);
            });
          }
          __syncthreads();   // B5: the column waves' partial sums
          store_irrep(d, W, io, pass);
          __syncthreads();   // B6
        }
      }
      if (a.executed && lane == 0) {                           // (profiling) 16 x 16 x 32 x 2 flops per matrix instruction: the hidden layer 14 per k tile and edge tile, step A 3 per (segment slot, tile)
        long long n_mfma = 2LL * 2 * KTP * 14 * NC;
        double useful = 144.0 * 144.0 * (len[0] + len[1]);
        for (int io = 0; io < W.n_io; ++io) {
          n_mfma += (long long)W.nct[io] * (3LL * CZ_NKT * CZ_MAXSEG * NC);
          useful += (double)W.nc_valid[io] * 145.0 * ((double)(len[0] + len[1]) + 48.0 * (nsg[0] + nsg[1]));
        }
        atomicAdd(a.executed, 16384.0 * (double)n_mfma);
        atomicAdd(a.executed + 1, 6.0 * useful);
        atomicAdd(a.executed + 2, 4.0 * 48.0 * W.n_io * (nsg[0] + nsg[1]));
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
      int ncb = 0;                                             // column blocks per group that hold a segment (the larger of the two groups')
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
          int t0 = 0, t1 = 0;
#pragma unroll
          for (int v = 0; v < NWA; ++v) { t0 += b_nseg[v]; t1 += b_nseg[NWA + v]; }
          ncb = __builtin_amdgcn_readfirstlane((max(t0, t1) + 15) >> 4);
        }
        for (int io = 0; io < W.n_io; ++io) {
          // step B: 9 per (column block, k-step, group, tile) -- of the k tile 9 one k-step, by column wave 0 only
          n_mfma += (long long)W.nct[io] * (9LL * ncb * NC * KTP * KS);
          f32x4 acc[3][NCB];
#pragma unroll
          for (int wt = 0; wt < 3; ++wt)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[wt][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const char* wbase = reinterpret_cast<const char*>(W.W2z) + (size_t)lane * 16 + (size_t)bw * (KS * CZ_TILE_BYTES);
          // W2z tile of (c tile ct, k tile 5 pass + jj): (ct0 + ct) 10 + 5 pass + jj
          auto tile_q = [&](int ct, int jj) { return (W.ct0[io] + ct) * CZ_NKT + KTP * pass + jj; };
          const int q_last = tile_q(W.nct[io] - 1, KTP - 1);
          // W2' fragments of my k-steps [tile parity][k-step][w tile][hi, lo]: this tile's, and the next tile's requested at the top of this one (what bounds the
          // tile loop is the stream of these fragments; it needs one tile in flight to run at its rate, profiles/TUNING_r6.md).  Five tiles per c tile: the parity
          // is that of the tile's index in a pair of c tiles, which is how the loop below is unrolled
          u32x4 Wf[2][KS][3][2];
          auto fetchW = [&](auto par_c, int q) {
            constexpr int par = decltype(par_c)::value;
            const char* p = wbase + (size_t)q * (8 * CZ_TILE_BYTES);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
              for (int wt = 0; wt < 3; ++wt)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) Wf[par][ks][wt][pc] = *reinterpret_cast<const u32x4*>(p + ks * CZ_TILE_BYTES + (wt * 2 + pc) * 1024);
          };
          // One PHASE: group gr of a tile -- column block by column block, k-step by k-step: the block's pieces (read one ahead), 9 matrix instructions (three
          // accumulators in turn)
          auto phase = [&](auto par_c, auto gr_c, auto ncb_c) {
            constexpr int par = decltype(par_c)::value, gr = decltype(gr_c)::value, NCBV = decltype(ncb_c)::value;
            const char* zr = zb + gr * CZ_GBYTES + (bw * KS) * CZ_VSTRIDE + g * 256 + n * 16;
            // (one set of piece registers: two fragment tiles and the accumulators of 64 columns leave no room for a second; the wave's other SIMD mates -- two chunk
            // waves -- fill the LDS round trip between two blocks)
            cz_static_for<0, KS * NCBV>([&](auto b_c) {
              constexpr int b = decltype(b_c)::value, ks = b / NCBV, cbl = b % NCBV, cb = GCB * gr + cbl;
              SL;
              const u32x4 ph = *reinterpret_cast<const u32x4*>(zr + ks * CZ_VSTRIDE + cbl * CZ_CB);
              const u32x4 pl = *reinterpret_cast<const u32x4*>(zr + ks * CZ_VSTRIDE + cbl * CZ_CB + CZ_PLANE);
              if constexpr ((ABL & 8) == 0) {
                const f16x8 zh = __builtin_bit_cast(f16x8, ph), zl = __builtin_bit_cast(f16x8, pl);
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[par][ks][wt][0]), zl, acc[wt][cb]);
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[par][ks][wt][1]), zh, acc[wt][cb]);
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[par][ks][wt][0]), zh, acc[wt][cb]);
              } else {
                asm volatile("" :: "v"(ph), "v"(pl));
              }
              SL;
            });
          };
          using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
          using G0 = std::integral_constant<int, 0>; using G1 = std::integral_constant<int, 1>;
          fetchW(P0{}, tile_q(0, 0));
          __syncthreads();   // B3
          // (the column-block count is decided once per unit, OUTSIDE the tile loops; c tiles in pairs: ten tiles, the fragment sets alternate)
          // c tiles in pairs (ten tiles: the fragment sets alternate and come back to set 0), a last single one if their number is odd -- no branch inside
          // the unrolled tiles (behind a skipped tile hipcc spills the fragments it has just requested)
          auto tiles = [&](auto ncb_c, auto n_c, int ct) {
            cz_static_for<0, decltype(n_c)::value * KTP>([&](auto t_c) {
              constexpr int t = decltype(t_c)::value, jj = t % KTP, par = t & 1;
              const int ctt = ct + t / KTP;
              stamp(10);
              const int qn = jj + 1 < KTP ? tile_q(ctt, jj + 1) : min(tile_q(ctt + 1, 0), q_last);
              fetchW(std::integral_constant<int, par ^ 1>{}, qn);   // (no branch around the fetch: the last one fetches a tile again)
              // (the k tile 9 -- the second pass's fifth: the bias row, only k_local 0 of k-step 0 holds anything -- goes through like any other: a branch for it
              // costs the registers more than its zeros cost the pipe)
              phase(std::integral_constant<int, par>{}, G0{}, ncb_c);
              stamp(11);
              __syncthreads();   // B4a
              phase(std::integral_constant<int, par>{}, G1{}, ncb_c);
              stamp(12);
              __syncthreads();   // B4b
              stamp(13);
            });
          };
          auto run_tiles = [&](auto ncb_c) {
            int ct = 0;
            for (; ct + 1 < W.nct[io]; ct += 2) tiles(ncb_c, std::integral_constant<int, 2>{}, ct);
            if (ct < W.nct[io]) tiles(ncb_c, std::integral_constant<int, 1>{}, ct);
          };
          run_tiles(std::integral_constant<int, GCB>{});         // (both column blocks of a group always: a second instantiation for units with <= 16 segments per group is 15 KB of code; its columns hold no segment and are not stored)
          // ---- the output irrep is complete for this pass: my partial sums (over my k-steps) to LDS; all twelve waves add and store
          float* red = reinterpret_cast<float*>(zb) + bw * (12 * NCB * 64);
#pragma unroll
          for (int wt = 0; wt < 3; ++wt)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
              for (int q = 0; q < 4; ++q) red[((wt * NCB + cb) * 4 + q) * 64 + lane] = acc[wt][cb][q];
          __syncthreads();   // B5
          store_irrep(d, W, io, pass);
          __syncthreads();   // B6
        }
      }
      if (a.executed && lane == 0) atomicAdd(a.executed, 16384.0 * (double)n_mfma);
    }
  }
#undef C_SH
#undef C_SA
#undef C_UA
#undef C_ROW
#undef C_SEG
#undef C_FIRST
#undef C_MASK
}
#undef MF
#undef SL

size_t convz_lds_bytes() { return CZ_ZBYTES + CZ_UNIT * CZ_CHUNK_FLOATS * sizeof(float) + CZ_BLOCK_INTS * sizeof(int); }

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
