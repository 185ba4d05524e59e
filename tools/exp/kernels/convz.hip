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

// Workgroup shape (round 6): TWELVE waves, three per SIMD, in two roles -- eight CHUNK waves (one chunk each: hidden layer, Y, step A, the cutting of Z into
// fp16 pieces) and four COLUMN waves (two k-steps of step B each: W2' fragments, accumulators).  Measured on MI355X (tools/exp/valu_rate.hip): a wave issues
// one vector instruction per ~5 cycles whatever its partner does, so what a SIMD gets done beside its matrix pipe grows with the number of waves on it;
// the symmetric eight-wave form of round 5 (every wave both roles, 256 registers, 148 values in scratch) left the pipe idle half of a tile, a four-wave
// form with 512 registers per wave (no scratch in the tile loop) was slower still (2 500 vs 1 900 cycles per tile).  Split by role neither wave needs
// more than 168 registers.
#define CZ_NWA 8                    // chunk waves = chunks per unit
#define CZ_NWB 4                    // column waves
#define CZ_NW (CZ_NWA + CZ_NWB)
#define CZ_KS (8 / CZ_NWB)          // k-steps of 32 (of the 256 (c, k) values of a tile) per column wave

// Z in LDS, already cut into fp16 pieces by the chunk wave that computed it.  Per buffer and column block two planes (hi, lo), each
// [k-step v: 1040 B][lane group g'': 256 B][column: 16 B = (k_local v: c 4g''..4g''+3 | k_local v + 8: the same c)].  Step A's lane (k_local n, group g) writes
// its four values c_local = 4 g + q as one 8-byte store per plane to (v = n & 7, g'' = g, half n >> 3); step B's lane (column, g'') reads the 16 bytes of its
// k-step from each plane: they ARE the B operand.  The 1040-byte stride (4 dwords mod 32 banks) makes the stores conflict-free, the reads are contiguous.
#define CZ_VSTRIDE 1040
#define CZ_PLANE (8 * CZ_VSTRIDE)
#define CZ_CB (2 * CZ_PLANE)        // one column block: hi plane, lo plane
#define CZ_NCB 2                    // column blocks of 16: the <= 8 x CZ_MAXSEG segments of the workgroup's eight chunks take consecutive columns
#define CZ_BUF (CZ_NCB * CZ_CB)     // one (c, k) tile: all column blocks
#define CZ_ZBYTES (2 * CZ_BUF)      // two buffers: step A of the next tile writes while step B of this one reads
#define CZ_WAVE_FLOATS (32 * 12 + 32 + 32 + 32 + 32 + 32 + CZ_MAXSEG * 16 + 24 * 64)   // per chunk wave: harmonics [32][12] | sa | ua | gather row offsets | segment ids | first slot of segment j | masks [CZ_MAXSEG segments][4 lane groups][4 dwords] | gathered x of the next c tile [8 slots x 3 components][64 lanes]
#ifndef CZ_PF_KT
#define CZ_PF_KT 5                  // the k tile behind which the next c tile's x gathers set out (used at k tile 9, behind an explicit wait)
#endif
#define CZ_ZSCALE (-20)             // |Z| <= 32 edges x 2^15 x 2^15 = 2^35 -> 2^15
// W1h (lin.0 in the tile format of conv2h.hip: 9 tiles of 9 one-KiB fragment pieces + 64 bytes of bias) is brought into LDS ONCE per unit by the column
// waves while the chunk waves set up their slots (every chunk wave fetching its own copy of the 83 KB from L2, two steps ahead of the MFMAs, was 60 k
// cycles of a unit's prologue): tiles 0..6 into the Z buffers (idle until the tile loops start), tiles 7 and 8 into a region of their own.
#define CZ_BLOCK_INTS (CZ_NWA + 2 * 16 * CZ_NCB + CZ_MAXCT * 16)                 // b_nseg | b_col_edge | b_col_inv | b_cdesc
#define CZ_W1X_OFF (CZ_ZBYTES + CZ_NWA * CZ_WAVE_FLOATS * 4 + ((CZ_BLOCK_INTS * 4 + 15) / 16) * 16)   // byte offset of W1h tiles 7, 8
__device__ __forceinline__ constexpr int cz_w1_lds(int m, int q) {   // LDS byte offset of piece q (0..7 fragments [piece][k-step], 8 tail, 9 bias) of W1h tile m
  return (m < 7 ? m * CH_TILE_BYTES : CZ_W1X_OFF + (m - 7) * CH_TILE_BYTES) + q * 1024;
}
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

// Both roles walk the same sequence of workgroup barriers (marked B1 .. B7 in both bodies): per unit B1 (the block-level arrays of the last unit are no longer
// read), B2 (segment counts of the eight chunks written), then per output irrep B3 (first Z tile written), one B4 per (c, k) tile, B5 (partial sums written), B6 (stored).
template <int ABL = 0>   // ABL (developer, wrong results): 4 no step A, 8 no step B, 32 prologue only; 128 (right results): s_memtime timeline of workgroup 0's first unit
__global__ __launch_bounds__(64 * CZ_NW) void k_convz(ConvZArgs a) {
  constexpr int KT = 9, NWA = CZ_NWA, NWB = CZ_NWB, KS = CZ_KS, NCB = CZ_NCB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave: a scalar register)
  const int n = lane & 15, g = lane >> 4;
  char* zb = reinterpret_cast<char*>(lds);
  int* b_nseg = reinterpret_cast<int*>(lds + CZ_ZBYTES / 4 + NWA * CZ_WAVE_FLOATS);   // [NWA]
  int* b_col_edge = b_nseg + NWA;                           // [16 NCB] message row of the column's segment, -1: column unused
  float* b_col_inv = reinterpret_cast<float*>(b_col_edge + 16 * NCB);   // [16 NCB] takes the chunk's factors off
  unsigned* b_cdesc = reinterpret_cast<unsigned*>(b_col_inv + 16 * NCB);   // [CZ_MAXCT][16] the conv's c-tile descriptors (a dependent global load per use otherwise)

  // developer timeline (ABL & 128, DBFR_CONVZ_DEBUG=<file>): s_memtime stamps of workgroup 0's first unit, [wave][stamp] unsigned long long in a.dbg
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
  // the output irrep is complete: add the column waves' partial sums (B5 behind them), take the factors off, store into the segments' first rows -- all twelve waves
  auto store_irrep = [&](const ConvZDesc& d, const ConvZ& W, int io, int ncb) {
    for (int idx = tid; idx < 16 * ncb * 48; idx += 64 * CZ_NW) {
      const int col = idx / 48, w = idx - 48 * col;
      const int e = b_col_edge[col];
      if (e >= 0) {
        const int r = w & 15;
        const float* rp = reinterpret_cast<const float*>(zb) + (((w >> 4) * NCB + (col >> 4)) * 4 + (r & 3)) * 64 + 16 * (r >> 2) + (col & 15);
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NWB; ++v) s += rp[v * (12 * NCB * 64)];
        d.msg[(size_t)e * d.D_out + W.out_off[io] + w] = s * b_col_inv[col] * W.rowinv[io * 48 + w];
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
    float* w_xs = reinterpret_cast<float*>(w_mask + CZ_MAXSEG * 16);   // [24][64] the next c tile's gathered x values: row = slot t (+ 8 m: component m of a vector input), one dword per lane
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
      if (ABL & 32) __syncthreads();   // B1 (developer variant that leaves a unit behind B2: nothing of the last unit still reads the block-level arrays; otherwise B6 of the last unit has seen to that)
      tr_on = (ABL & 128) && blockIdx.x == 0 && unit == 2 * (int)gridDim.x && (a.dbg_sel == 0 || a.dbg_sel == a.c[0].w.ct0[a.c[0].w.n_io - 1] + a.c[0].w.nct[a.c[0].w.n_io - 1]);
      if (tr_on) tr_n = 0;
      stamp(1);
      // ---- every load the prologue needs that depends on the chunk's first edge only, requested in one go (under the W2' stream of 256 workgroups a dependent
      // round trip to memory costs 2 000 - 20 000 cycles: the prologue has two of them -- this one, and the gathered radial-MLP rows behind the indices)
      // slots: lanes 0..31 own slot L (clamped to the chunk's last edge beyond its length; a chunk without edges reads edge 0 of a non-empty conv or nothing)
      const int sl = lane & 31;
      const bool have = len > 0;
      const int e_sl = have ? es + min(sl, len - 1) : 0;
      int tgt_l = 0, gth_l = 0;
      float shv[9];
      float xmx = 0.f;
      f32x4 Ba[2][KT];                                         // raw radial-MLP inputs of my edge of both edge tiles: [emb | tab1 row | tab2 row], 16 floats of each per lane group
      int e_et[2], i1_et[2], i2_et[2];
      {
        if (have) { tgt_l = d.tgt[e_sl]; gth_l = d.gth[e_sl]; }
#pragma unroll
        for (int et = 0; et < 2; ++et) {
          e_et[et] = have ? es + min(16 * et + n, len - 1) : 0;
          i1_et[et] = have ? d.idx1[e_et[et]] : 0;
          i2_et[et] = have ? d.idx2[e_et[et]] : 0;
        }
        const float* sp = d.sh + (size_t)e_sl * SH_LD;
#pragma unroll
        for (int k = 0; k < 9; ++k) shv[k] = (have && lane < 32) ? sp[k] : 0.f;
        if (have && d.xmax) xmx = d.xmax[gidx];               // per graph, computed once per layer (k_row_absmax): no pass over the gathered rows here
#pragma unroll
        for (int et = 0; et < 2; ++et) {
          const float* r0 = d.emb + (size_t)e_et[et] * NS;
#pragma unroll
          for (int s4 = 0; s4 < 3; ++s4) Ba[et][s4] = have ? *reinterpret_cast<const f32x4*>(r0 + 16 * s4 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int et = 0; et < 2; ++et) {
          const float* r1 = d.tab1 + (size_t)i1_et[et] * d.ld1;
          const float* r2 = d.tab2 + (size_t)i2_et[et] * d.ld2;
#pragma unroll
          for (int s4 = 0; s4 < 3; ++s4) {
            Ba[et][3 + s4] = have ? *reinterpret_cast<const f32x4*>(r1 + 16 * s4 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
            Ba[et][6 + s4] = have ? *reinterpret_cast<const f32x4*>(r2 + 16 * s4 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      const int tgt_prev = __shfl_up(tgt_l, 1);
      const bool is_first = lane < 32 && sl < len && (sl == 0 || tgt_l != tgt_prev);
      const unsigned firsts = (unsigned)__ballot(is_first);
      const int nseg = min(__popc(firsts), CZ_MAXSEG);         // (the chunk table holds no chunk with more)
      const int seg_l = (sl < len) ? __popc(firsts & (0xffffffffu >> (31 - sl))) - 1 : -1;
      stamp(20);
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
      // ---- bounds for the y scale: largest |x| over the gathered rows, largest |harmonic|
      if (have && !d.xmax) {
        const float* xr = d.x + (size_t)gth_l * d.ldx;
        const int d4 = d.ldx >> 2;
        for (int j = lane >> 5; j < d4; j += 2) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * j);
          xmx = fmaxf(fmaxf(xmx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
      }
      stamp(21);
      xmx = cz_wave_max(xmx); smx = cz_wave_max(smx);
      int ey = 0;                                              // y is multiplied by 2^ey: |y| <= 3 |x| |sh| < 2^(ex + es + 2) -> below 2^15
      if (xmx > 0.f && smx > 0.f) ey = max(-100, min(100, 13 - __builtin_amdgcn_frexp_expf(xmx) - __builtin_amdgcn_frexp_expf(smx)));
      ey = __builtin_amdgcn_readfirstlane(ey);
      const float sY = __builtin_amdgcn_ldexpf(1.f, ey);
      const int nseg_u = __builtin_amdgcn_readfirstlane(nseg);

      stamp(2);
      // Y of a c tile, MASKED per segment: lane (c = n, group g), my eight slots; Ym[j] = the pieces with the slots outside segment j zeroed -- the
      // A operand of step A for all ten k tiles of the c tile.  The gathers of a c tile are requested several tiles before they are used
      // (prefetch_x -> finish_Y).
      u32x4 Ym[CZ_MAXSEG][2];
#pragma unroll
      for (int j = 0; j < CZ_MAXSEG; ++j) { Ym[j][0] = (u32x4){0u, 0u, 0u, 0u}; Ym[j][1] = (u32x4){0u, 0u, 0u, 0u}; }
      // The gathers go STRAIGHT TO LDS (global_load_lds_dword: no register holds them), a few tiles before the c tile starts:
      // row t (+ 8 m for component m of a vector input) of the wave's stage, one dword per lane.
      // (inline assembly on purpose: for the builtin hipcc's wait-count pass makes EVERY later LDS read wait for vmcnt(0); here the one wait sits in
      // front of the stage's readers, wait_stage)
      // (m0 is written without being declared clobbered: hipcc treats it as a reserved register and rejects the clobber; nothing the compiler emits for
      // this kernel uses m0 -- no movrel indexing, no LDS-DMA builtin, no GWS -- which diffbindfr_amd/build.py checks on the generated assembly)
      auto prefetch_x = [&](int ctg) {   // ctg: c tile of the conv (over both irreps)
        if (nseg_u == 0) return;
        const unsigned cd_n = b_cdesc[ctg * 16 + n];
        const bool vec = (__builtin_amdgcn_readfirstlane(cd_n) >> 12) & 1u;
        const int xo = cd_n & 0xfff;
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
      // every load issued so far has landed -- the stage's gathers among them (a chunk wave has no other loads in flight in the tile loop)
      auto wait_stage = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      };
      auto finish_Y = [&](int ctg) {
        if (nseg_u == 0) return;
        const unsigned cd_n = b_cdesc[ctg * 16 + n];
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
      // ---- hidden layer, transposed: D[edge, unit] = sum_f a[edge, f] W1[unit, f]; A = the edge's inputs (cut per edge), B = W1h tiles from LDS.
      // The two edge tiles one after the other (inputs of 16 edges in registers at a time: with both, the layer's inputs, results and fragment ring
      // exceed the 168 registers of a wave and spill); the fragments come from LDS, so the second walk over W1h costs LDS reads only.
      u32x4 Hh[CZ_NKT][2];                                     // H pieces [k tile][hi, lo]: lane (unit n, group g), eight edges {4g..4g+3, 16+4g..16+4g+3}
      int eh = 0;                                              // the factor on h is 2^(15 - ehc + k1) = phi, also what the constant 1 of the bias becomes
      {
        float Hf[KT][2][4];
        float hmx = 0.f;
        float sa_[2];
#pragma unroll
        for (int et = 0; et < 2; ++et) {
          const int slot = 16 * et + n;
          float amx = 0.f;
#pragma unroll
          for (int s4 = 0; s4 < KT; ++s4)
#pragma unroll
            for (int q = 0; q < 4; ++q) amx = fmaxf(amx, fabsf(Ba[et][s4][q]));
          amx = fmaxf(amx, __shfl_xor(amx, 16));
          amx = fmaxf(amx, __shfl_xor(amx, 32));
          const int ja = max(-14, __builtin_amdgcn_frexp_expf(amx));
          sa_[et] = __builtin_amdgcn_ldexpf(1.f, 15 - ja);
          if (g == 0) { w_sa[slot] = sa_[et]; w_ua[slot] = __builtin_amdgcn_ldexpf(1.f, ja - 15); }
        }
        __syncthreads();   // B1b: W1h is in LDS, and the c-tile descriptors (column waves)
        stamp(4);
        prefetch_x(0);                                         // the first c tile's gathers travel behind the hidden layer
        cz_static_for<0, 2>([&](auto et_c) {
          constexpr int et = decltype(et_c)::value;
          u32x4 Ah[2][4];                                      // input pieces [hi, lo][k-step of 32]
          u32x4 Atc;                                           // last 16 k: [lo | hi]
#pragma unroll
          for (int s4 = 0; s4 < KT; ++s4) {
            unsigned hi0, lo0, hi1, lo1;
            cz_split2(Ba[et][s4][0] * sa_[et], Ba[et][s4][1] * sa_[et], hi0, lo0);
            cz_split2(Ba[et][s4][2] * sa_[et], Ba[et][s4][3] * sa_[et], hi1, lo1);
            if (s4 < 8) {
              Ah[0][s4 >> 1][2 * (s4 & 1)] = hi0; Ah[0][s4 >> 1][2 * (s4 & 1) + 1] = hi1;
              Ah[1][s4 >> 1][2 * (s4 & 1)] = lo0; Ah[1][s4 >> 1][2 * (s4 & 1) + 1] = lo1;
            } else {
              Atc = (u32x4){lo0, lo1, hi0, hi1};
            }
          }
          // my rows of a result tile are the edges 16 et + 4 g + q: their factors
          const f32x4 sar = *reinterpret_cast<const f32x4*>(w_sa + 16 * et + 4 * g), uar = *reinterpret_cast<const f32x4*>(w_ua + 16 * et + 4 * g);
          // W1 fragments: step i = 5 m + s (s < 4: the k-step of 32 of tile m, hi and lo piece; s = 4: the tile's last 16 k [hi | lo] and, in the second
          // register, my unit's bias) through a ring of three register pairs, read TWO steps ahead of their MFMAs
          u32x4 F[3][2];
          auto ldF = [&](auto ic) {
            constexpr int i = decltype(ic)::value, m = i / 5, sx = i % 5;
            if constexpr (sx < 4) {
              F[i % 3][0] = *reinterpret_cast<const u32x4*>(zb + cz_w1_lds(m, sx) + lane * 16);
              F[i % 3][1] = *reinterpret_cast<const u32x4*>(zb + cz_w1_lds(m, 4 + sx) + lane * 16);
            } else {
              F[i % 3][0] = *reinterpret_cast<const u32x4*>(zb + cz_w1_lds(m, 8) + lane * 16);
              F[i % 3][1][0] = *reinterpret_cast<const unsigned*>(zb + cz_w1_lds(m, 9) + n * 4);
            }
          };
          ldF(std::integral_constant<int, 0>{});
          ldF(std::integral_constant<int, 1>{});
          // (tile m's bias arrives with its LAST step; the accumulator starts from zero and the bias x the edge's factor is added behind the tile)
          f32x4 acc;
          cz_static_for<0, 5 * KT>([&](auto ic) {
            constexpr int i = decltype(ic)::value, m = i / 5, sx = i % 5;
            SL;                                                // (pins the step order: left to itself hipcc hoists the LDS reads of many steps and spills)
            if constexpr (i + 2 < 5 * KT) ldF(std::integral_constant<int, (i + 2 < 5 * KT ? i + 2 : 0)>{});
            if constexpr (sx == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (sx < 4) {
              const f16x8 whi = __builtin_bit_cast(f16x8, F[i % 3][0]), wlo = __builtin_bit_cast(f16x8, F[i % 3][1]);
              acc = MF(__builtin_bit_cast(f16x8, Ah[1][sx]), whi, acc);
              acc = MF(__builtin_bit_cast(f16x8, Ah[0][sx]), wlo, acc);
              acc = MF(__builtin_bit_cast(f16x8, Ah[0][sx]), whi, acc);
            } else {
              // k = 128..143: W tail fragment = [hi (4) | lo (4)], input tail = [lo | hi]: one x32 MFMA carries both small products; the large product
              // on the x32 instruction too, upper half of the operands zero (an x16 MFMA that takes as SrcC an accumulator an x32 MFMA has just
              // written reads stale data on MI355X + ROCm 7.2 -- DESIGN.md section 4, invariant 3 -- and 18 half-empty MFMAs per chunk cost nothing)
              const u32x4 wt = F[i % 3][0];
              const f16x8 wf = __builtin_bit_cast(f16x8, wt);
              const f16x8 wh = __builtin_bit_cast(f16x8, (u32x4){wt[0], wt[1], 0u, 0u});
              const float bias = __builtin_bit_cast(float, F[i % 3][1][0]);
              const float w1r = W.W1rinv ? W.W1rinv[16 * m + n] : 1.f;
              acc = MF(__builtin_bit_cast(f16x8, Atc), wf, acc);
              acc = MF(__builtin_bit_cast(f16x8, (u32x4){Atc[2], Atc[3], 0u, 0u}), wh, acc);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float v = fmaxf(acc[q] + bias * sar[q], 0.f) * (uar[q] * w1r);   // = 2^k1 h (w1r: my unit's row factor of W1h off)
                Hf[m][et][q] = v;
                hmx = fmaxf(hmx, v);
              }
            }
          });
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
        hmx = cz_wave_max(hmx);
        // 2^k1 h < 2^eh; the factor 2^(15 - ehc) on it makes phi = 2^(15 - ehc + k1) on h: ehc >= k1 keeps phi, the image of the bias's 1, inside fp16
        eh = max(hmx > 0.f ? __builtin_amdgcn_frexp_expf(hmx) : W.k1, W.k1);
        eh = __builtin_amdgcn_readfirstlane(min(eh, W.k1 + 100));
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
      __syncthreads();   // B2: the segment counts of the eight chunks; b_cdesc and the cleared column table (column waves)
      // ---- columns: the segments of the eight chunks side by side (<= 8 x CZ_MAXSEG = 32 = two column blocks of step B)
      int cbase = 0, total = 0;
#pragma unroll
      for (int v = 0; v < NWA; ++v) { const int cnt = b_nseg[v]; if (v < wave) cbase += cnt; total += cnt; }
      cbase = __builtin_amdgcn_readfirstlane(cbase); total = __builtin_amdgcn_readfirstlane(total);   // (wave-uniform: scalar branches below)
      const int ncb = (total + 15) >> 4;                       // column blocks of 16 in step B: 0 (no edges), 1 or 2
      const float zs = __builtin_amdgcn_ldexpf(1.f, CZ_ZSCALE);
      if (ABL & 32) continue;
      if (lane < nseg_u) {
        b_col_edge[cbase + lane] = es + w_first[lane];
        b_col_inv[cbase + lane] = __builtin_amdgcn_ldexpf(1.f, -CZ_ZSCALE - ey - ephi);
      }
      // LDS addresses of my segments' Z columns in buffer 0, hi plane (lane part + column).  A slot without a segment computes zeros and writes them into
      // column 31: unused whenever such a slot exists (the unit then has fewer than 32 segments), and never stored
      char* za[CZ_MAXSEG];
      {
        char* zw0 = zb + (n & 7) * CZ_VSTRIDE + g * 256 + (n >> 3) * 8;
#pragma unroll
        for (int j = 0; j < CZ_MAXSEG; ++j) {
          const int col = cbase + j;
          za[j] = j < nseg_u ? zw0 + (col >> 4) * CZ_CB + (col & 15) * 16 : zw0 + (NCB - 1) * CZ_CB + 15 * 16;
        }
      }
      // (profiling) matrix instructions this wave issues in this unit, and the flops of them that are not padding: the hidden layer ...
      long long n_mfma = 2LL * KT * 14;
      double useful = 3.0 * 2.0 * 144.0 * 144.0 * len;
      for (int io = 0; io < W.n_io; ++io) {
        // ... step A: 3 per (segment slot, tile); not padding: the edges of a segment x (valid (path, u) pairs) x 145 (k tile 9: one column).  Step B is counted
        // by the column waves -- except that each chunk wave adds what is not padding there for its own segments: one column per segment x (c, k) values x 48
        n_mfma += (long long)W.nct[io] * (3LL * CZ_NKT * CZ_MAXSEG);
        useful += 3.0 * 2.0 * (double)W.nc_valid[io] * 145.0 * ((double)len + 48.0 * nseg_u);
#define YM(j, p) __builtin_bit_cast(f16x8, Ym[j][p])
        // step A of one (c, k) tile: Z[c, k] of my four segment slots -- independent three-product chains side by side -- x 2^-20 (|Z| <= 32 x 2^15 x 2^15), cut
        // into two fp16 pieces and stored (column = segment): the B operand of step B as the column waves read it
        auto stepA = [&](auto kt_c, int boff) {
          constexpr int kt = decltype(kt_c)::value;
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
          for (int j = 0; j < CZ_MAXSEG; ++j) {
            unsigned h0, l0, h1, l1;
            cz_split2(z[j][0] * zs, z[j][1] * zs, h0, l0);
            cz_split2(z[j][2] * zs, z[j][3] * zs, h1, l1);
            *reinterpret_cast<u32x2*>(za[j] + boff) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(za[j] + boff + CZ_PLANE) = (u32x2){l0, l1};
          }
        };
#undef YM
        // While step B reads tile i from buffer i & 1, step A of tile i + 1 is written into the other buffer; one barrier per tile.
        const int ct0 = W.ct0[io], n_ct = W.ct0[W.n_io - 1] + W.nct[W.n_io - 1];
        wait_stage();                                          // (requested behind B1b / behind tile CZ_PF_KT of the irrep before)
        finish_Y(ct0);
        stepA(std::integral_constant<int, 0>{}, 0);
        __syncthreads();   // B3
        for (int ct = 0; ct < W.nct[io]; ++ct) {
          cz_static_for<0, CZ_NKT>([&](auto kt_c) {
            constexpr int kt = decltype(kt_c)::value;
            stamp(10);
            if constexpr (kt < KT) {
              stepA(std::integral_constant<int, kt + 1>{}, ((kt + 1) & 1) * CZ_BUF);
              // (the next c tile's gathers set out here; used at kt = 9)
              if constexpr (kt == CZ_PF_KT) if (ct0 + ct + 1 < n_ct) prefetch_x(ct0 + ct + 1);   // (the last c tile of an irrep: the first of the next)
            } else {
              // the k tile 9 is being read by column wave 0: the next c tile's Y and its first step A
              if (ct + 1 < W.nct[io]) { wait_stage(); finish_Y(ct0 + ct + 1); stepA(std::integral_constant<int, 0>{}, 0); }
            }
            stamp(12);
            __syncthreads();   // B4: the next tile's Z is complete; this tile's buffer may be written again
            stamp(13);
          });
        }
        __syncthreads();   // B5: the column waves' partial sums
        store_irrep(d, W, io, ncb);
        __syncthreads();   // B6
      }
      if (a.executed && lane == 0) {                           // 16 x 16 x 32 x 2 flops per instruction
        atomicAdd(a.executed, 16384.0 * (double)n_mfma);
        atomicAdd(a.executed + 1, useful);
        atomicAdd(a.executed + 2, 4.0 * 48.0 * W.n_io * nseg_u);   // (bytes of the form that runs: a segment's scalar-output columns are written once)
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
      if (ABL & 32) __syncthreads();   // B1 (see the chunk waves)
      tr_on = (ABL & 128) && blockIdx.x == 0 && unit == 2 * (int)gridDim.x && (a.dbg_sel == 0 || a.dbg_sel == a.c[0].w.ct0[a.c[0].w.n_io - 1] + a.c[0].w.nct[a.c[0].w.n_io - 1]);
      if (tr_on) tr_n = 0;
      stamp(1);
      __builtin_amdgcn_s_sleep(4);                             // (the chunk waves' first requests -- the head of the prologue's only chain of dependent loads -- ahead of the 92 KB requested here)
      {                                                        // W1h into LDS: 90 pieces (1 KiB; the bias piece 64 B) dealt round the four waves -- all requests of a wave in flight at
        // once, through registers (as LDS-DMA, global_load_lds_dwordx4 with a new m0 each, the 23 requests of a wave took 48 k cycles under the W2' stream:
        // one after the other)
        const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)W.W1h, 0, KT * CH_TILE_BYTES, 0x00020000);
        constexpr int NP = (10 * KT + NWB - 1) / NWB;
        u32x4 buf[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const int i = min(bw + NWB * j, 10 * KT - 1), m = i / 10, q = i - 10 * m;
          buf[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rW1, lane * 16, m * CH_TILE_BYTES + q * 1024, 0));
        }
        // the block-level tables
        const int t = tid - 64 * NWA;                          // 0 .. 255
        if (t < CZ_MAXCT * 16) b_cdesc[t] = t < (W.ct0[W.n_io - 1] + W.nct[W.n_io - 1]) * 16 ? W.cdesc[t] : 0u;
        if (t < 16 * NCB) b_col_edge[t] = -1;
        stamp(30);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const int i = bw + NWB * j, m = i / 10, q = i - 10 * m;
          if (i < 10 * KT && (q < 9 || lane < 4))                // (the bias piece is 64 bytes)
            *reinterpret_cast<u32x4*>(zb + (m < 7 ? m * CH_TILE_BYTES : CZ_W1X_OFF + (m - 7) * CH_TILE_BYTES) + q * 1024 + lane * 16) = buf[j];
        }
        stamp(31);
      }
      __syncthreads();   // B1b: W1h is in LDS
      stamp(32);
      __syncthreads();   // B2
      stamp(33);
      int total = 0;
#pragma unroll
      for (int v = 0; v < NWA; ++v) total += b_nseg[v];
      total = __builtin_amdgcn_readfirstlane(total);
      const int ncb = (total + 15) >> 4;
      if (ABL & 32) continue;
      long long n_mfma = 0;
      for (int io = 0; io < W.n_io; ++io) {
        // step B: 9 per (column block, k-step, tile) -- of the k tile 9 one k-step, by column wave 0 only
        n_mfma += (long long)W.nct[io] * (9LL * ncb * (KT * KS + (bw == 0)));
        f32x4 acc[3][NCB];
#pragma unroll
        for (int wt = 0; wt < 3; ++wt)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[wt][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* wbase = reinterpret_cast<const char*>(W.W2z) + (size_t)lane * 16 + (size_t)bw * (KS * CZ_TILE_BYTES);
        int gq = W.ct0[io] * CZ_NKT;                           // running (c tile, k tile) index into W2z; CZ_NKT is even: the Z buffer of tile gq is kt & 1
        const int gq_last = (W.ct0[io] + W.nct[io]) * CZ_NKT - 1;
        // W2' fragments of my k-steps: [tile parity][k-step][w tile][hi, lo] -- a ring of TWO tiles in which a fragment pair is requested again, for the tile after
        // next, as soon as its last matrix instruction has been issued.  What bounds the tile loop is the stream of these fragments (48 KB per tile and
        // workgroup out of L2, ~1 800 cycles of latency under the load of 256 such streams): bytes per cycle = bytes in flight / latency, and with the whole next
        // tile requested at the top of a tile (rounds 5 and 6 until here) one tile was in flight -- 26 B per cycle and workgroup, whatever the waves did meanwhile.
        u32x4 Wf[2][KS][3][2];
        auto fetchG = [&](auto par_c, auto ks_c, auto wt_c, int q) {   // one fragment pair of tile q
          constexpr int par = decltype(par_c)::value, ks = decltype(ks_c)::value, wt = decltype(wt_c)::value;
          const char* p = wbase + (size_t)q * (8 * CZ_TILE_BYTES) + ks * CZ_TILE_BYTES + wt * 2048;
          Wf[par][ks][wt][0] = *reinterpret_cast<const u32x4*>(p);
          Wf[par][ks][wt][1] = *reinterpret_cast<const u32x4*>(p + 1024);
        };
        auto fetchW = [&](auto par_c, int q) {                 // a whole tile (no branch around these loads: behind a conditional fetch hipcc's wait-count pass assumes nothing about what is in flight)
          cz_static_for<0, KS * 3>([&](auto f_c) {
            constexpr int f = decltype(f_c)::value;
            fetchG(par_c, std::integral_constant<int, f / 3>{}, std::integral_constant<int, f % 3>{}, q);
          });
        };
        using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
        // One tile kt < 9: the Z pieces of my k-steps, then the KS x 3 fragment pairs two at a time -- their 6 NCBV matrix instructions interleaved (W_hi Z_lo,
        // W_lo Z_hi, W_hi Z_hi per column block; an accumulator comes round every fourth / second instruction), then the two pairs of the tile after next
        auto tileB = [&](auto kt_c, auto ncb_c) {
          constexpr int kt = decltype(kt_c)::value, NCBV = decltype(ncb_c)::value, par = kt & 1;
          const char* zr = zb + par * CZ_BUF + (bw * KS) * CZ_VSTRIDE + g * 256 + n * 16;
          u32x4 ph[KS][NCBV], pl[KS][NCBV];
          SL;
#pragma unroll
          for (int cb = 0; cb < NCBV; ++cb) {
            ph[0][cb] = *reinterpret_cast<const u32x4*>(zr + cb * CZ_CB);
            pl[0][cb] = *reinterpret_cast<const u32x4*>(zr + cb * CZ_CB + CZ_PLANE);
          }
          SL;
          const int q2 = min(gq + 2, gq_last);
          cz_static_for<0, (KS * 3) / 2>([&](auto pp_c) {
            constexpr int pp = decltype(pp_c)::value, fa = 2 * pp, fb = 2 * pp + 1, ksa = fa / 3, wta = fa % 3, ksb = fb / 3, wtb = fb % 3;
            if constexpr (pp == 0 && KS > 1) {                 // the pieces of the next k-step, behind the first instructions of this one
#pragma unroll
              for (int cb = 0; cb < NCBV; ++cb) {
                ph[1][cb] = *reinterpret_cast<const u32x4*>(zr + CZ_VSTRIDE + cb * CZ_CB);
                pl[1][cb] = *reinterpret_cast<const u32x4*>(zr + CZ_VSTRIDE + cb * CZ_CB + CZ_PLANE);
              }
            }
            if constexpr ((ABL & 8) == 0) {
              cz_static_for<0, 3>([&](auto pr_c) {
                constexpr int pr = decltype(pr_c)::value;
#pragma unroll
                for (int cb = 0; cb < NCBV; ++cb) {
                  const f16x8 za_ = __builtin_bit_cast(f16x8, pr == 0 ? pl[ksa][cb] : ph[ksa][cb]), zb_ = __builtin_bit_cast(f16x8, pr == 0 ? pl[ksb][cb] : ph[ksb][cb]);
                  acc[wta][cb] = MF(__builtin_bit_cast(f16x8, Wf[par][ksa][wta][pr == 1]), za_, acc[wta][cb]);
                  acc[wtb][cb] = MF(__builtin_bit_cast(f16x8, Wf[par][ksb][wtb][pr == 1]), zb_, acc[wtb][cb]);
                }
              });
            }
            SL;
            fetchG(std::integral_constant<int, par>{}, std::integral_constant<int, ksa>{}, std::integral_constant<int, wta>{}, q2);
            fetchG(std::integral_constant<int, par>{}, std::integral_constant<int, ksb>{}, std::integral_constant<int, wtb>{}, q2);
            SL;
          });
          if constexpr ((ABL & 8) != 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
              for (int cb = 0; cb < NCBV; ++cb) asm volatile("" :: "v"(ph[ks][cb]), "v"(pl[ks][cb]));
          }
        };
        // the k tile 9 (the bias row: k-step 0 only, column wave 0's first); every wave's fragments of this tile make way for those of the tile after next
        auto tile9 = [&]() {
          constexpr int kt = KT;
          if (bw == 0 && !(ABL & 8)) {
            const char* zr = zb + (kt & 1) * CZ_BUF + g * 256 + n * 16;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
              if (cb < ncb) {
                const f16x8 zh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(zr + cb * CZ_CB));
                const f16x8 zl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(zr + cb * CZ_CB + CZ_PLANE));
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[1][0][wt][0]), zl, acc[wt][cb]);
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[1][0][wt][1]), zh, acc[wt][cb]);
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) acc[wt][cb] = MF(__builtin_bit_cast(f16x8, Wf[1][0][wt][0]), zh, acc[wt][cb]);
              }
          }
          SL;
          fetchW(K1{}, min(gq + 2, gq_last));
        };
        fetchW(K0{}, gq);
        fetchW(K1{}, min(gq + 1, gq_last));
        __syncthreads();   // B3
        // (the column-block count is decided once per unit, OUTSIDE the tile loops)
        auto run_tiles = [&](auto ncb_c) {
          for (int ct = 0; ct < W.nct[io]; ++ct) {
            cz_static_for<0, CZ_NKT>([&](auto kt_c) {
              constexpr int kt = decltype(kt_c)::value;
              stamp(10);
              if constexpr (kt < KT) tileB(kt_c, ncb_c); else tile9();
              stamp(12);
              ++gq;
              __syncthreads();   // B4
              stamp(13);
            });
          }
        };
        if (ncb > 1) run_tiles(std::integral_constant<int, 2>{}); else run_tiles(std::integral_constant<int, 1>{});
        // ---- the output irrep is complete: my partial sums (over my k-steps) to LDS; all twelve waves add and store
        float* red = reinterpret_cast<float*>(zb) + bw * (12 * NCB * 64);
#pragma unroll
        for (int wt = 0; wt < 3; ++wt)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[((wt * NCB + cb) * 4 + q) * 64 + lane] = acc[wt][cb][q];
        __syncthreads();   // B5
        store_irrep(d, W, io, ncb);
        __syncthreads();   // B6
      }
      if (a.executed && lane == 0) atomicAdd(a.executed, 16384.0 * (double)n_mfma);
    }
  }
}
#undef MF
#undef SL

size_t convz_lds_bytes() { return CZ_W1X_OFF + 2 * CH_TILE_BYTES; }

void launch_convz(const ConvZArgs& a0, hipStream_t st) {
  ConvZArgs a = a0;
  static float* dbg_dev = nullptr;
  static const char* dbg_file = getenv("DBFR_CONVZ_DEBUG");
  if (dbg_file && !dbg_dev) {
    if (hipMalloc(&dbg_dev, 12 * 512 * 8) != hipSuccess || hipMemset(dbg_dev, 0, 12 * 512 * 8) != hipSuccess) dbg_dev = nullptr;
    else atexit([] {
      std::vector<float> h(12 * 512 * 2);
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
  if (abl == 4) V(4) if (abl == 8) V(8) if (abl == 12) V(12) if (abl == 32) V(32) if (abl == 128) V(128)
#endif
  V(0)
#undef V
}
