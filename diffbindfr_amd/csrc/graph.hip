// Per-step graph construction and embeddings for gfx950.
//
// Replaces (every reverse-diffusion step, positions move):
//   torch_cluster.radius_graph / radius          tpscore.py:586,613,655-660,721-723,747-749
//   get_complete_bipartite_graph                 torch_utils/graph.py:81-140 (lig x {CA,CB})
//   edge_vec, GaussianSmearing, o3.spherical_harmonics, FullTensorProduct(sh,"2e")
//                                                tpscore.py:593-598,615-620,676-680,704-708,715-729,740-755
//   SimpleLinear edge/node embeddings, AtomEncoder, sinusoidal_embedding   tpscore.py:464-479
//
// Layout: one workgroup per (graph, edge set); the graph's coordinates sit in LDS, one thread per
// scatter-target node scans its candidates in index order (N <= ~900 per graph: brute force in LDS
// beats any cell list).  Edge lists come out grouped by target node (CSR: row_start/row_cnt), graphs
// in batch order, so that downstream reductions are contiguous and reproducible:
//   k_edges_count -> k_edges_scan (per set, one block) -> k_edges_fill.
// torch_cluster semantics reproduced: strict d^2 < r^2, "first max_num_neighbors by index" per query,
// radius_graph queries max+1 then drops the self loop.
#include "common.h"

#define MAX_NL 256      // per-graph ligand atoms (the ligand kernels of heads.hip keep a conformer in static LDS)
#define MAX_NA 8192     // per-graph pocket atoms: the graph's coordinates are staged in dynamic LDS sized by the batch's
                        // largest graph (16 B per atom, 128 KB of the CU's 160 KB at the limit)

enum SetKind { SET_LL = 0, SET_AA = 1, SET_AL = 2, SET_LA = 3, SET_TOR = 4, SET_SC = 5, N_SETS = 6 };

struct GraphArgs {
  dbfr_batch b;
  const int* lig_batch;   // [NL]
  const int* atm_batch;   // [NA]
  const uint8_t* is_cab;  // [NA]
  const int* n_cab;       // [G]
  const float* tr_sigma;  // [G]
  float lig_cut2, atom_cut2, cross_cut2;
  int lig_cap, atom_cap, dynamic_cross;
  EdgeSet set[N_SETS];
  int* err;               // device status block: [0] status bits, [1] first step (+1) whose edge lists overflowed,
                          // [2+k] set k overflowed in THIS step, [8+k] largest edge count seen for set k
  int step;               // sampler step index (0 for dbfr_score)
  int lds_nl, lds_na;     // LDS staging capacities (>= max_nl / max_na of the batch)
  int n_chunk;            // EDGE_CHUNK-target chunks per graph every set is split into (own workgroup each)
  int lanes;              // lanes per target of the edge builder (1 | 4); n_chunk and lanes are chosen ONCE per call (dbfr_edge_form) and
                          // shared by every launcher below: the g_cnt / g_base layout depends on both
};

__device__ __forceinline__ float d2_rn(float ax, float ay, float az, float bx, float by, float bz) {
  // ((dx*dx + dy*dy) + dz*dz) with every op rounded (no fma contraction): the oracle's order
  float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ void sh_l2(float x, float y, float z, float* o) {
  // component-normalised real harmonics l=2, y polar (e3nn): sqrt15 xz, sqrt15 xy, sqrt5(y^2-(x^2+z^2)/2), ...
  const float s15 = 3.872983346207417f, s5 = 2.23606797749979f;
  o[0] = s15 * x * z;
  o[1] = s15 * x * y;
  o[2] = s5 * (y * y - 0.5f * (x * x + z * z));
  o[3] = s15 * y * z;
  o[4] = (s15 * 0.5f) * (z * z - x * x);
}

__device__ __forceinline__ float vec_sh(float vx, float vy, float vz, float* sh9) {
  float nrm = sqrtf(vx * vx + vy * vy + vz * vz);
  float inv = 1.0f / fmaxf(nrm, 1e-12f);
  float x = vx * inv, y = vy * inv, z = vz * inv;
  const float s3 = 1.7320508075688772f;
  sh9[0] = 1.0f;
  sh9[1] = s3 * x;
  sh9[2] = s3 * y;
  sh9[3] = s3 * z;
  sh_l2(x, y, z, sh9 + 4);
  return nrm;
}

// shared staging of one graph (views into the dynamic LDS block)
struct GraphLds {
  float *lx, *ly, *lz;
  float *ax, *ay, *az;
  int* thr;               // radius_graph cap threshold per centre
  int* scan;              // [256]
};

__device__ __forceinline__ void lds_views(GraphLds& s, float* base, int nl, int na) {
  s.scan = reinterpret_cast<int*>(base);
  s.lx = base + 256; s.ly = s.lx + nl; s.lz = s.ly + nl;
  s.ax = s.lz + nl; s.ay = s.ax + na; s.az = s.ay + na;
  s.thr = reinterpret_cast<int*>(s.az + na);      // max(nl, na) entries
}
static size_t lds_bytes(int nl, int na) { return 4 * (256 + 3 * (size_t)nl + 3 * (size_t)na + (size_t)(na > nl ? na : nl)); }

__device__ void load_graph(GraphLds& s, const GraphArgs& A, int g, int kind, int& l0, int& nl, int& a0, int& na) {
  l0 = A.b.lig_ptr[g];
  nl = A.b.lig_ptr[g + 1] - l0;
  a0 = A.b.atm_ptr[g];
  na = A.b.atm_ptr[g + 1] - a0;
  float scale = 1.0f;
  bool scaled = (kind == SET_AL || kind == SET_LA) && A.dynamic_cross;
  if (scaled) scale = __fadd_rn(__fmul_rn(A.tr_sigma[g], 0.2f), 5.0f);
  if (kind != SET_AA && kind != SET_SC)
    for (int i = threadIdx.x; i < nl; i += blockDim.x) {
      float x = A.b.lig_pos[3 * (l0 + i)], y = A.b.lig_pos[3 * (l0 + i) + 1], z = A.b.lig_pos[3 * (l0 + i) + 2];
      if (scaled) { x = __fdiv_rn(x, scale); y = __fdiv_rn(y, scale); z = __fdiv_rn(z, scale); }
      s.lx[i] = x; s.ly[i] = y; s.lz[i] = z;
    }
  if (kind == SET_AA || kind == SET_AL || kind == SET_LA || kind == SET_SC)
    for (int i = threadIdx.x; i < na; i += blockDim.x) {
      float x = A.b.rec_pos[3 * (a0 + i)], y = A.b.rec_pos[3 * (a0 + i) + 1], z = A.b.rec_pos[3 * (a0 + i) + 2];
      if (scaled) { x = __fdiv_rn(x, scale); y = __fdiv_rn(y, scale); z = __fdiv_rn(z, scale); }
      s.ax[i] = x; s.ay[i] = y; s.az[i] = z;
    }
  __syncthreads();
}

// radius_graph cap: centre i keeps the first (cap+1) in-range points incl. itself, by index
__device__ void cap_thresholds(GraphLds& s, const float* px, const float* py, const float* pz, int n, float r2,
                               int cap) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int thr = 0x7fffffff;
    if (n > cap + 1) {
      int c = 0;
      for (int j = 0; j < n; ++j)
        if (d2_rn(px[i], py[i], pz[i], px[j], py[j], pz[j]) < r2 && ++c == cap + 1) { thr = j; break; }
    }
    s.thr[i] = thr;
  }
  __syncthreads();
}

// Count (EMIT == false) or emit the edges of target `t` (local index) of the given set whose gather candidates lie in the lane's
// share [lo, hi) of the candidate range (EDGE_LANES lanes per target, contiguous shares in index order; the lane with `first` also
// takes the ligand's bond edges, which come first in the reference's edge list).  Returns the number of edges of the share.
// `base` = absolute slot of the share's first edge.
template <bool EMIT>
__device__ int target_edges(const GraphLds& s, const GraphArgs& A, int kind, int g, int t, int l0, int nl, int a0,
                            int na, int lo, int hi, bool first, int base) {
  const EdgeSet& S = A.set[kind];
  int cnt = 0;
  auto emit = [&](int tgt, int gth, int aux, float vx, float vy, float vz) {
    if (EMIT) {
      int e = base + cnt;
      if (e < S.cap) {
        float sh9[9];
        float d = vec_sh(vx, vy, vz, sh9);
        S.tgt[e] = tgt; S.gth[e] = gth; S.aux[e] = aux; S.dist[e] = d;
#pragma unroll
        for (int k = 0; k < 9; ++k) S.sh[(size_t)e * SH_LD + k] = sh9[k];
      }
    }
    ++cnt;
  };
  const float* LP = A.b.lig_pos;
  const float* RP = A.b.rec_pos;
  if (kind == SET_LL) {
    const int gt = l0 + t;
    if (first)
      for (int k = A.b.bond_ptr[gt]; k < A.b.bond_ptr[gt + 1]; ++k) {  // bond edges (u=t -> v): tgt=u, gth=v
        int v = A.b.bond_dst[k];
        emit(gt, v, k, LP[3 * v] - LP[3 * gt], LP[3 * v + 1] - LP[3 * gt + 1], LP[3 * v + 2] - LP[3 * gt + 2]);
      }
    for (int i = lo; i < hi; ++i) {
      if (i == t || t > s.thr[i]) continue;
      if (d2_rn(s.lx[i], s.ly[i], s.lz[i], s.lx[t], s.ly[t], s.lz[t]) < A.lig_cut2) {
        int gi = l0 + i;  // radius edge (neighbour=t, centre=i): scatter to t, gather from i
        emit(gt, gi, -1, LP[3 * gi] - LP[3 * gt], LP[3 * gi + 1] - LP[3 * gt + 1], LP[3 * gi + 2] - LP[3 * gt + 2]);
      }
    }
  } else if (kind == SET_AA) {
    const int gt = a0 + t;
    for (int i = lo; i < hi; ++i) {
      if (i == t || t > s.thr[i]) continue;
      if (d2_rn(s.ax[i], s.ay[i], s.az[i], s.ax[t], s.ay[t], s.az[t]) < A.atom_cut2) {
        int gi = a0 + i;
        emit(gt, gi, -1, RP[3 * gi] - RP[3 * gt], RP[3 * gi + 1] - RP[3 * gt + 1], RP[3 * gi + 2] - RP[3 * gt + 2]);
      }
    }
  } else if (kind == SET_AL) {  // target ligand atom, gathers pocket atoms; vec = rec - lig
    const int gt = l0 + t;
    for (int i = lo; i < hi; ++i) {
      int gi = a0 + i;
      bool in = A.is_cab[gi] || d2_rn(s.lx[t], s.ly[t], s.lz[t], s.ax[i], s.ay[i], s.az[i]) < A.cross_cut2;
      if (in) emit(gt, gi, -1, RP[3 * gi] - LP[3 * gt], RP[3 * gi + 1] - LP[3 * gt + 1], RP[3 * gi + 2] - LP[3 * gt + 2]);
    }
  } else if (kind == SET_LA) {  // target pocket atom, gathers ligand atoms; harmonics of the SAME vec = rec - lig
    const int gt = a0 + t;
    const bool cab = A.is_cab[gt];
    for (int i = lo; i < hi; ++i) {
      int gi = l0 + i;
      bool in = cab || d2_rn(s.lx[i], s.ly[i], s.lz[i], s.ax[t], s.ay[t], s.az[t]) < A.cross_cut2;
      if (in) emit(gt, gi, -1, RP[3 * gt] - LP[3 * gi], RP[3 * gt + 1] - LP[3 * gi + 1], RP[3 * gt + 2] - LP[3 * gi + 2]);
    }
  }
  return cnt;
}

// pseudotorque graphs: target = torsion bond (mid-point), gathers atoms within r, first `cap` by index.  The lane handles the
// candidates [lo, hi) and keeps at most `keep` of them (what the cap leaves for its share once the lanes before it took theirs).
template <bool EMIT>
__device__ int torsion_edges(const GraphArgs& A, int kind, int tgt, const float* P, int p0, const float* px,
                             const float* py, const float* pz, int u, int v, float r2, int lo, int hi, int keep, int base) {
  const EdgeSet& S = A.set[kind];
  float mx = (P[3 * u] + P[3 * v]) / 2, my = (P[3 * u + 1] + P[3 * v + 1]) / 2, mz = (P[3 * u + 2] + P[3 * v + 2]) / 2;
  float b9[9];
  if (EMIT) vec_sh(P[3 * v] - P[3 * u], P[3 * v + 1] - P[3 * u + 1], P[3 * v + 2] - P[3 * u + 2], b9);
  int cnt = 0;
  for (int i = lo; i < hi && cnt < keep; ++i) {
    if (d2_rn(mx, my, mz, px[i], py[i], pz[i]) < r2) {
      if (EMIT) {
        int e = base + cnt;
        if (e < S.cap) {
          int gi = p0 + i;
          float a9[9];
          float d = vec_sh(P[3 * gi] - mx, P[3 * gi + 1] - my, P[3 * gi + 2] - mz, a9);
          S.tgt[e] = tgt; S.gth[e] = gi; S.aux[e] = -1; S.dist[e] = d;
          // FullTensorProduct(sh(0e+1o+2e), 2e): only the 0e (2e x 2e), 1o (1o x 2e), 1e (2e x 2e) blocks
          // can reach the scalar outputs of the torsion conv; coefficient sqrt(2 l_out + 1) folded.
          const float* a1 = a9 + 1; const float* a2 = a9 + 4; const float* b = b9 + 4;
          const float r3 = 1.7320508075688772f;
          float* o = S.sh + (size_t)e * SH_LD;
          o[0] = (a2[0] * b[0] + a2[1] * b[1] + a2[2] * b[2] + a2[3] * b[3] + a2[4] * b[4]) * 0.4472135954999579f;
          {  // 1o = sqrt3 * w3j(1,2,1): symmetric traceless M(b) a1 / sqrt10
            float m00 = -b[2] - r3 * b[4], m01 = r3 * b[1], m02 = r3 * b[0], m11 = 2.f * b[2], m12 = r3 * b[3],
                  m22 = -b[2] + r3 * b[4];
            const float c = 0.31622776601683794f;
            o[1] = c * (m00 * a1[0] + m01 * a1[1] + m02 * a1[2]);
            o[2] = c * (m01 * a1[0] + m11 * a1[1] + m12 * a1[2]);
            o[3] = c * (m02 * a1[0] + m12 * a1[1] + m22 * a1[2]);
          }
          {  // 1e = sqrt3 * w3j(2,2,1)[i,j,k] a2_i b_j ; table in units of 1/sqrt30
            const float c = 0.31622776601683794f;
            float k0 = -a2[0] * b[1] + a2[1] * b[0] + r3 * a2[2] * b[3] - r3 * a2[3] * b[2] + a2[3] * b[4] - a2[4] * b[3];
            float k1 = -2.f * a2[0] * b[4] - a2[1] * b[3] + a2[3] * b[1] + 2.f * a2[4] * b[0];
            float k2 = a2[0] * b[3] + r3 * a2[1] * b[2] + a2[1] * b[4] - r3 * a2[2] * b[1] - a2[3] * b[0] - a2[4] * b[1];
            o[4] = c * k0; o[5] = c * k1; o[6] = c * k2;
          }
          o[7] = 0.f; o[8] = 0.f;
        }
      }
      ++cnt;
    }
  }
  return cnt;
}

__device__ __forceinline__ void target_range(const GraphArgs& A, int kind, int g, int& t0, int& nt) {
  const int* ptr = (kind == SET_LL || kind == SET_AL) ? A.b.lig_ptr
                 : (kind == SET_AA || kind == SET_LA) ? A.b.atm_ptr
                 : (kind == SET_TOR)                  ? A.b.tor_ptr
                                                      : A.b.sc_ptr;
  t0 = ptr[g];
  nt = ptr[g + 1] - t0;
}

// Work split (round 4).  A workgroup serves EDGE_CHUNK = 64 consecutive targets of one (graph, set) with EDGE_LANES = 4 lanes per
// target: lane q of a target walks the q-th quarter of the candidate range, so the per-thread serial walk of a 866-atom pocket is
// 217 candidates instead of 866 and a graph is 4 x as many workgroups (BASELINE config 1 -- ONE pocket x 4 poses -- was 16 + 16
// workgroups per step on a 256-CU chip, 9 % of its latency).  Thread order = (target, quarter) = emission order, so the block scan
// over the per-LANE counts yields every lane's first slot directly and the edge list is the one the one-lane walk produced: by
// target, bonds first, then by gather index.  The "first cap by index" rule of the pseudotorque sets is applied on the quarter
// prefix: a lane keeps what the cap leaves once the lanes before it took theirs.
// EDGE_LANES = 1 (a workgroup = 256 targets, the round-1 form) is kept for batches that fill the chip anyway: there the quartered walk
// buys nothing and four times as many workgroups each stage the graph and run the block scan (measured at 640 poses: 310.7 -> 296.7
// poses/s with four lanes everywhere); launch_edges picks by the number of workgroups.  Both forms emit the same list.
template <bool EMIT, int EDGE_LANES>
__global__ __launch_bounds__(256) void k_edges(GraphArgs A) {
  constexpr int TPB = 256 / EDGE_LANES;              // targets per pass
  extern __shared__ float dyn_lds[];
  GraphLds s;
  lds_views(s, dyn_lds, A.lds_nl, A.lds_na);
  const int C = A.n_chunk;
  const int g = blockIdx.x / C, chunk = blockIdx.x - g * C, kind = blockIdx.y;
  const EdgeSet& S = A.set[kind];
  if (S.cap == 0) return;
  // this set did not fit its capacity in this step: leave it EMPTY (row_cnt 0) so that nothing downstream reads past
  // the buffers; the host re-plans with the counted sizes and resumes from this step (dbfr_capacity_report)
  const bool overflow = EMIT && A.err[2 + kind] != 0;
  int t0, nt;  // first global target id of the graph, number of targets
  target_range(A, kind, g, t0, nt);
  // chunk c serves targets [c TPB, (c + 1) TPB); the last chunk takes whatever lies beyond (launch_edges sizes C by the
  // batch's largest pocket / ligand, which bounds every set's targets per graph: chi angles < side-chain atoms, torsions < atoms)
  const int c_begin = chunk * TPB, c_end = chunk == C - 1 ? nt : min(nt, c_begin + TPB);
  if (c_begin >= nt) { if (!EMIT && threadIdx.x == 0) S.g_cnt[blockIdx.x] = 0; return; }
  int l0, nl, a0, na;
  load_graph(s, A, g, kind, l0, nl, a0, na);
  if (kind == SET_LL) cap_thresholds(s, s.lx, s.ly, s.lz, nl, A.lig_cut2, A.lig_cap);
  if (kind == SET_AA) cap_thresholds(s, s.ax, s.ay, s.az, na, A.atom_cut2, A.atom_cap);
  const int q = threadIdx.x & (EDGE_LANES - 1), tl = threadIdx.x / EDGE_LANES;
  const int lane = threadIdx.x & 63, quad0 = lane & ~(EDGE_LANES - 1);
  // gather candidates of the set: ligand atoms (LL, LA, TOR) or pocket atoms (AA, AL, SC)
  const int ncand = (kind == SET_LL || kind == SET_LA || kind == SET_TOR) ? nl : na;
  const int lo = (int)(((long)ncand * q) / EDGE_LANES), hi = (int)(((long)ncand * (q + 1)) / EDGE_LANES);
  int running = EMIT ? S.g_base[blockIdx.x] : 0;
  int total = 0;
  for (int c0 = c_begin; c0 < c_end; c0 += TPB) {  // (one pass, but for the last chunk of an oversized set)
    const int t = c0 + tl;
    const bool live = t < c_end && !overflow;
    int cnt = 0, keep = 0x7fffffff, tu = 0, tv = 0;
    if (live) {
      const bool known = EMIT && EDGE_LANES == 1;   // one lane per target: the count pass left the target's count in row_cnt
      if (kind == SET_TOR) {
        const int k = A.b.tor_bond[t0 + t];
        tu = A.b.bond_src[k]; tv = A.b.bond_dst[k];
      } else if (kind == SET_SC) {
        // side-chain bonds: mid-point of (j,k) vs the graph's pocket atoms; s.ax holds UNSCALED coords for SC
        tu = A.b.sc_bond[2 * (t0 + t)]; tv = A.b.sc_bond[2 * (t0 + t) + 1];
      }
      if (known) cnt = S.row_cnt[t0 + t];
      else if (kind <= SET_LA) cnt = target_edges<false>(s, A, kind, g, t, l0, nl, a0, na, lo, hi, q == 0, 0);
      else if (kind == SET_TOR) cnt = torsion_edges<false>(A, kind, t0 + t, A.b.lig_pos, l0, s.lx, s.ly, s.lz, tu, tv, A.lig_cut2, lo, hi, A.lig_cap, 0);
      else cnt = torsion_edges<false>(A, kind, t0 + t, A.b.rec_pos, a0, s.ax, s.ay, s.az, tu, tv, A.atom_cut2, lo, hi, A.lig_cap, 0);
    }
    if (kind > SET_LA && EDGE_LANES > 1) {   // first `cap` by index: what the lanes before this one leave (all four lanes of a quad take this path together)
      int before = 0;
#pragma unroll
      for (int j = 0; j < EDGE_LANES - 1; ++j) { const int cj = __shfl(cnt, quad0 + j); if (j < q) before += cj; }
      keep = max(0, A.lig_cap - before);
      cnt = min(cnt, keep);
    }
    // inclusive scan over the 256 threads = over (target, quarter) in emission order
    s.scan[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      int v = threadIdx.x >= o ? s.scan[threadIdx.x - o] : 0;
      __syncthreads();
      s.scan[threadIdx.x] += v;
      __syncthreads();
    }
    const int incl = s.scan[threadIdx.x], pass_total = s.scan[255];
    __syncthreads();
    if (t < c_end) {
      int tot = cnt;      // the target's edge count = sum over its lanes
      if (EDGE_LANES > 1) { tot += __shfl_xor(tot, 1); tot += __shfl_xor(tot, 2); }
      static_assert(EDGE_LANES == 1 || EDGE_LANES == 4, "lane sums are written for 1 or 4 lanes per target");
      if (kind > SET_LA && EDGE_LANES == 1) keep = cnt;
      const int base = overflow ? 0 : running + incl - cnt;
      if (!EMIT) {
        if (q == 0) S.row_cnt[t0 + t] = tot;
      } else {
        if (q == 0) { S.row_start[t0 + t] = base; if (overflow) S.row_cnt[t0 + t] = 0; }
        if (overflow || cnt == 0) {
        } else if (kind <= SET_LA) {
          target_edges<true>(s, A, kind, g, t, l0, nl, a0, na, lo, hi, q == 0, base);
        } else if (kind == SET_TOR) {
          torsion_edges<true>(A, kind, t0 + t, A.b.lig_pos, l0, s.lx, s.ly, s.lz, tu, tv, A.lig_cut2, lo, hi, keep, base);
        } else {
          torsion_edges<true>(A, kind, t0 + t, A.b.rec_pos, a0, s.ax, s.ay, s.az, tu, tv, A.atom_cut2, lo, hi, keep, base);
        }
      }
    }
    running += pass_total;
    total += pass_total;
  }
  if (!EMIT && threadIdx.x == 0) S.g_cnt[blockIdx.x] = total;
}

// exclusive scan of the per-(graph, chunk) totals; one block per edge set
__global__ __launch_bounds__(256) void k_edges_scan(GraphArgs A) {
  __shared__ int sc[256];
  const EdgeSet& S = A.set[blockIdx.x];
  if (S.cap == 0) { if (threadIdx.x == 0 && S.n_edges) *S.n_edges = 0; return; }
  const int G = A.b.G * A.n_chunk;
  int running = 0;
  for (int c0 = 0; c0 < G; c0 += 256) {
    int g = c0 + threadIdx.x;
    int v = g < G ? S.g_cnt[g] : 0;
    sc[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      int w = threadIdx.x >= o ? sc[threadIdx.x - o] : 0;
      __syncthreads();
      sc[threadIdx.x] += w;
      __syncthreads();
    }
    if (g < G) S.g_base[g] = running + sc[threadIdx.x] - v;
    running += sc[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicMax(&A.err[8 + blockIdx.x], running);
    const bool over = running > S.cap;
    A.err[2 + blockIdx.x] = over;
    if (over) { atomicOr(A.err, 1); atomicCAS(&A.err[1], 0, A.step + 1); running = 0; }
    *S.n_edges = running;
  }
}

// workgroups per (graph, set) of the finer form: what plan() sizes g_cnt / g_base for (the coarser form uses a prefix of them)
int dbfr_edge_chunks(int max_na, int max_nl) { const int m = max_na > max_nl ? max_na : max_nl; return (m + EDGE_CHUNK - 1) / EDGE_CHUNK; }
static int edge_lanes(const dbfr_batch& b);
static int edge_chunks_of(const dbfr_batch& b) {      // workgroups per (graph, set) of the form launch_edges picks for this batch
  const int m = b.max_na > b.max_nl ? b.max_na : b.max_nl, tpb = 256 / edge_lanes(b);
  return (m + tpb - 1) / tpb;
}
static int edge_lanes(const dbfr_batch& b) {
  static const int forced = getenv("DBFR_EDGE_LANES") ? atoi(getenv("DBFR_EDGE_LANES")) : 0;      // developer: 1 | 4
  if (forced == 1 || forced == 4) return forced;
  const int m = b.max_na > b.max_nl ? b.max_na : b.max_nl;
  const long wg1 = (long)b.G * ((m + 255) / 256);          // workgroups per set with one lane per target
  return wg1 >= 2L * dbfr_current_cu_count() ? 1 : 4;      // four lanes while the coarse form would leave the chip mostly empty
}

void dbfr_edge_form(const dbfr_batch& b, int* n_chunk, int* lanes) { *lanes = edge_lanes(b); *n_chunk = edge_chunks_of(b); }

void launch_edges(const GraphArgs& A0, bool with_heads_only, hipStream_t st) {
  (void)with_heads_only;
  GraphArgs A = A0;
  A.lds_nl = (A.b.max_nl + 63) & ~63;
  A.lds_na = (A.b.max_na + 63) & ~63;
  const size_t lds = lds_bytes(A.lds_nl, A.lds_na);
  if (lds > 64 * 1024) {   // above the default 64 KB a kernel has to be told -- per DEVICE, and a process may drive several: no
    // "done once" flag; the call is cheap next to a launch that stages a > 4 k-atom pocket
    for (const void* f : {reinterpret_cast<const void*>(&k_edges<false, 1>), reinterpret_cast<const void*>(&k_edges<true, 1>),
                          reinterpret_cast<const void*>(&k_edges<false, 4>), reinterpret_cast<const void*>(&k_edges<true, 4>)})
      if (dbfr_launch_check(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64), "k_edges: hipFuncSetAttribute(MaxDynamicSharedMemorySize)")) return;
  }
  if (A.n_chunk <= 0) dbfr_edge_form(A.b, &A.n_chunk, &A.lanes);      // n_chunk <= what plan() in api.cpp sized g_cnt / g_base for (dbfr_edge_chunks)
  if (A.lanes == 1) {
    hipLaunchKernelGGL((k_edges<false, 1>), dim3(A.b.G * A.n_chunk, N_SETS), dim3(256), lds, st, A);
    hipLaunchKernelGGL(k_edges_scan, dim3(N_SETS), dim3(256), 0, st, A);
    hipLaunchKernelGGL((k_edges<true, 1>), dim3(A.b.G * A.n_chunk, N_SETS), dim3(256), lds, st, A);
  } else {
    hipLaunchKernelGGL((k_edges<false, 4>), dim3(A.b.G * A.n_chunk, N_SETS), dim3(256), lds, st, A);
    hipLaunchKernelGGL(k_edges_scan, dim3(N_SETS), dim3(256), 0, st, A);
    hipLaunchKernelGGL((k_edges<true, 4>), dim3(A.b.G * A.n_chunk, N_SETS), dim3(256), lds, st, A);
  }
}

// Chunks for the reduce-first conv (convz.hip): consecutive edges of ONE graph, at most 32 of them and at most CZ_MAXSEG scatter targets (a chunk
// never holds edges of two graphs, and where it is cut depends on the graph's own targets only: which edges are summed together does not depend
// on batch mates).  chunk_len: the next chunk of the edge range [es, hi), called by a whole wave; the length is wave-uniform.
__device__ __forceinline__ int chunk_len(const int* tgt, int es, int hi, unsigned* starts = nullptr) {
  const int lane = threadIdx.x & 63;
  const int n = min(32, hi - es);
  int t = 0;
  if (lane < n) t = tgt[es + lane];
  const int tp = __shfl_up(t, 1);
  unsigned f = (unsigned)__ballot(lane < n && (lane == 0 || t != tp));   // starts of the maximal runs of one target
  if (starts) *starts = f;
#pragma unroll
  for (int i = 0; i < CZ_MAXSEG; ++i) f &= f - 1;
  return f ? __ffs((int)f) - 1 : n;                      // up to the start of run CZ_MAXSEG + 1
}
// The walk over the chunks of [lo, hi) with ONE load per TWO chunks: 64 targets from the current edge on -- wherever the first chunk ends, the 32
// targets the second one is cut from lie among them.  (A dependent L2 round trip per chunk was most of what k_chunk_count / k_chunk_fill did: 300 us
// per step for four 3DBS-sized graphs, 8 % of such a call; now 225.  Streaming the targets through four registers of 64 with no round trip at all in
// the loop was built too and is SLOWER, 270 us: two ds_bpermute + select per chunk on top of the shuffle / ballot / ffs chain that every cut needs.)
// f(es, len, starts) -> false stops the walk.  Same chunks as chunk_len, by construction.
template <typename F>
__device__ __forceinline__ void chunk_walk(const int* tgt, int lo, int hi, F&& f) {
  const int lane = threadIdx.x & 63;
  auto cut = [&](int t, int n, unsigned& starts) {          // chunk_len on targets that are in registers (lanes < n <= 32)
    const int tp = __shfl_up(t, 1);
    unsigned fl = (unsigned)__ballot(lane < n && (lane == 0 || t != tp));
    starts = fl;
#pragma unroll
    for (int i = 0; i < CZ_MAXSEG; ++i) fl &= fl - 1;
    return fl ? __ffs((int)fl) - 1 : n;
  };
  for (int es = lo; es < hi;) {
    const int nl = min(64, hi - es);
    const int t = lane < nl ? tgt[es + lane] : 0;
    unsigned st;
    const int l1 = cut(t, min(32, nl), st);
    if (!f(es, l1, st)) return;
    es += l1;
    if (es >= hi) return;
    const int t2 = __shfl(t, (lane + l1) & 63);             // (l1 + min(32, hi - es) <= nl: inside the 64 loaded)
    const int l2 = cut(t2, min(32, hi - es), st);
    if (!f(es, l2, st)) return;
    es += l2;
  }
}
__device__ __forceinline__ void graph_edge_range(const GraphArgs& A, const EdgeSet& S, int g, int& lo, int& hi) {
  const int E = min(*S.n_edges, S.cap);
  lo = min(S.g_base[g * A.n_chunk], E);
  hi = g + 1 < A.b.G ? min(S.g_base[(g + 1) * A.n_chunk], E) : E;
}
// pass 1, one wave per (graph, set): the graph's first edge and its number of chunks (chunk0 holds the COUNT until k_graph_chunks scans it)
__global__ __launch_bounds__(256) void k_chunk_count(GraphArgs A) {
  const EdgeSet& S = A.set[blockIdx.y];
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= A.b.G) return;
  int lo = 0, cnt = 0;
  if (S.cap > 0) {
    int hi;
    graph_edge_range(A, S, g, lo, hi);
    chunk_walk(S.tgt, lo, hi, [&](int, int, unsigned) { ++cnt; return true; });
  }
  if ((threadIdx.x & 63) == 0) { S.chunk0[g] = cnt; S.gedge0[g] = lo; }
}
// pass 2, one block per set: chunk0[g] = number of chunks of the graphs before g; chunk0[G] = their total (k_convz's unit count)
__global__ __launch_bounds__(256) void k_graph_chunks(GraphArgs A) {
  __shared__ int sc[256];
  const EdgeSet& S = A.set[blockIdx.x];
  const int G = A.b.G;
  int running = 0;
  for (int g0 = 0; g0 < G; g0 += 256) {
    const int g = g0 + threadIdx.x;
    const int v = g < G ? S.chunk0[g] : 0;
    sc[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int w = threadIdx.x >= o ? sc[threadIdx.x - o] : 0;
      __syncthreads();
      sc[threadIdx.x] += w;
      __syncthreads();
    }
    if (g < G) S.chunk0[g] = running + sc[threadIdx.x] - v;
    running += sc[255];
    __syncthreads();
  }
  // (chunk_cap = cap / 32 + targets / 4 + G + 8 bounds the total: a chunk ends after 32 edges, at the end of its graph, or with the end of its
  // fourth target -- and every target ends once)
  if (threadIdx.x == 0) { S.chunk0[G] = min(running, S.chunk_cap); S.gedge0[G] = S.cap > 0 ? min(*S.n_edges, S.cap) : 0; }
}
// pass 3, one wave per (graph, set): the chunk table
__global__ __launch_bounds__(256) void k_chunk_fill(GraphArgs A) {
  const EdgeSet& S = A.set[blockIdx.y];
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= A.b.G || S.cap == 0) return;
  int lo, hi;
  graph_edge_range(A, S, g, lo, hi);
  int ch = S.chunk0[g];
  const int lane = threadIdx.x & 63;
  chunk_walk(S.tgt, lo, hi, [&](int es, int len, unsigned starts) {
    if (ch >= S.chunk_cap) return false;
    if (lane == 0) { S.chunk_es[ch] = es; S.chunk_gl[ch] = (g << 6) | len; }
    if (lane < len) S.seg_first[es + lane] = (starts >> lane) & 1u;      // (every edge lies in exactly one chunk: no clearing pass)
    ++ch;
    return true;
  });
}

void launch_graph_chunks(const GraphArgs& A0, hipStream_t st) {
  GraphArgs A = A0;
  if (A.n_chunk <= 0) dbfr_edge_form(A.b, &A.n_chunk, &A.lanes);
  hipLaunchKernelGGL(k_chunk_count, dim3((A.b.G + 3) / 4, N_SETS), dim3(256), 0, st, A);
  hipLaunchKernelGGL(k_graph_chunks, dim3(N_SETS), dim3(256), 0, st, A);
  hipLaunchKernelGGL(k_chunk_fill, dim3((A.b.G + 3) / 4, N_SETS), dim3(256), 0, st, A);
}

// The same table for ONE flat edge list (the single-conv test hooks, api.cpp test_conv_impl): the list is cut every `span` edges as if those were
// graphs, so that the walk runs in parallel.  cnt0 [n_span + 1] scratch -> first chunk of every span, total in cnt0[n_span].
__global__ __launch_bounds__(256) void k_flat_chunk_count(const int* tgt, const int* n_edges, int max_edges, int span, int n_span, int* cnt0) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_span) return;
  const int E = min(*n_edges, max_edges);
  const int lo = min(g * span, E), hi = min(lo + span, E);
  int cnt = 0;
  for (int es = lo; es < hi; ++cnt) es += chunk_len(tgt, es, hi);
  if ((threadIdx.x & 63) == 0) cnt0[g] = cnt;
}
__global__ __launch_bounds__(256) void k_flat_chunk_scan(int n_span, int* cnt0, int cap) {
  __shared__ int sc[256];
  int running = 0;
  for (int g0 = 0; g0 < n_span; g0 += 256) {
    const int g = g0 + threadIdx.x;
    const int v = g < n_span ? cnt0[g] : 0;
    sc[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int w = threadIdx.x >= o ? sc[threadIdx.x - o] : 0;
      __syncthreads();
      sc[threadIdx.x] += w;
      __syncthreads();
    }
    if (g < n_span) cnt0[g] = running + sc[threadIdx.x] - v;
    running += sc[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt0[n_span] = min(running, cap);
}
__global__ __launch_bounds__(256) void k_flat_chunk_fill(const int* tgt, const int* n_edges, int max_edges, int span, int n_span, const int* cnt0, int cap,
                                                         int* chunk_es, int* chunk_gl) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_span) return;
  const int E = min(*n_edges, max_edges);
  const int lo = min(g * span, E), hi = min(lo + span, E);
  int ch = cnt0[g];
  for (int es = lo; es < hi && ch < cap; ++ch) {
    const int len = chunk_len(tgt, es, hi);
    if ((threadIdx.x & 63) == 0) { chunk_es[ch] = es; chunk_gl[ch] = len; }   // (graph 0: the hooks pass no per-graph bounds)
    es += len;
  }
}
void launch_flat_chunks(const int* tgt, const int* n_edges, int max_edges, int span, int n_span, int* cnt0, int cap, int* chunk_es, int* chunk_gl, hipStream_t st) {
  hipLaunchKernelGGL(k_flat_chunk_count, dim3((n_span + 3) / 4), dim3(256), 0, st, tgt, n_edges, max_edges, span, n_span, cnt0);
  hipLaunchKernelGGL(k_flat_chunk_scan, dim3(1), dim3(256), 0, st, n_span, cnt0, cap);
  hipLaunchKernelGGL(k_flat_chunk_fill, dim3((n_span + 3) / 4), dim3(256), 0, st, tgt, n_edges, max_edges, span, n_span, cnt0, cap, chunk_es, chunk_gl);
}

// Largest |feature| over the node rows [ptr[g], ptr[g + 1]) of every graph: the bound k_convz scales its y operand with.  Per GRAPH, not per
// batch: a power of two derived from it must not depend on batch mates.  One workgroup per graph.
// (four independent loads per thread and trip, and 1 024 threads where the graphs are large: a 3DBS-sized pocket behind ONE workgroup of 256 threads
// with one load in flight took 26 us, a tenth of a 4-graph call's time.  A maximum does not depend on the order it is taken in.)
__global__ __launch_bounds__(1024) void k_row_absmax(const float* x0, const int* ptr0, float* out0, const float* x1, const int* ptr1, float* out1, int ld, int D, int G) {
  __shared__ float sm[16];
  const int NT = blockDim.x;
  const bool second = (int)blockIdx.x >= G;                  // blocks [0, G): the first node set (ligand), [G, 2 G): the second (pocket atoms)
  const int g = second ? blockIdx.x - G : blockIdx.x;
  const float* x = second ? x1 : x0;
  const int* ptr = second ? ptr1 : ptr0;
  const int r0 = ptr[g], r1 = ptr[g + 1];
  const int d4 = D >> 2;
  const long n = (long)(r1 - r0) * d4;
  auto at = [&](long i) {
    const int r = (int)(i / d4), c = (int)(i - (long)r * d4);
    return *reinterpret_cast<const float4*>(x + (size_t)(r0 + r) * ld + 4 * c);
  };
  auto amax = [](float m, const float4& v) { return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); };
  float m = 0.f;
  long i = threadIdx.x;
  for (; i + 3 * NT < n; i += 4 * NT) {
    const float4 v0 = at(i), v1 = at(i + NT), v2 = at(i + 2 * NT), v3 = at(i + 3 * NT);
    m = amax(amax(amax(amax(m, v0), v1), v2), v3);
  }
  for (; i < n; i += NT) m = amax(m, at(i));
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 16) {
    m = (int)threadIdx.x < (NT >> 6) ? sm[threadIdx.x] : 0.f;
    for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) (second ? out1 : out0)[g] = m;
  }
}
// both node sets of a layer in one launch (two launches per layer were 2-3 % of a 16-graph call)
void launch_row_absmax(const float* lx, const int* lig_ptr, float* out_l, const float* ax, const int* atm_ptr, float* out_a, int ld, int D, int G, int n_rows_a, hipStream_t st) {
  if (G <= 0) return;
  const int nt = (long)(n_rows_a / G) * (D >> 2) >= 16384 ? 1024 : 256;      // (mean float4s per graph of the larger set: ~64 trips of 256 threads and up)
  hipLaunchKernelGGL(k_row_absmax, dim3(2 * G), dim3(nt), 0, st, lx, lig_ptr, out_l, ax, atm_ptr, out_a, ld, D, G);
}

// dbfr_model_set_edge_log: per-graph edge counts of this step, log[k * G + g] = sum over the graph's target chunks
__global__ void k_edge_log(GraphArgs A, int* log, int stride) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (g >= A.b.G) return;
  const EdgeSet& S = A.set[k];
  int c = 0;
  if (S.cap > 0)
    for (int i = 0; i < A.n_chunk; ++i) c += S.g_cnt[g * A.n_chunk + i];
  log[k * stride + g] = c;
}

void launch_edge_log(const GraphArgs& A0, int* log_row, int stride, hipStream_t st) {
  GraphArgs A = A0;
  if (A.n_chunk <= 0) dbfr_edge_form(A.b, &A.n_chunk, &A.lanes);
  hipLaunchKernelGGL(k_edge_log, dim3((A.b.G + 255) / 256, N_SETS), dim3(256), 0, st, A, log_row, stride);
}

// Diagnostic (dbfr_model_set_tie_log; never on the data path): per graph and edge set, the number of candidate pairs whose distance lies within `tol` of
// the set's hard cutoff -- the pairs at which two runs that differ by rounding may build different graphs (the reference's radius graphs have no soft
// edge).  Distances as the edge builder forms them (d2_rn of the staged coordinates, the cross sets in units of the graph's dynamic cutoff, pseudotorque
// sets from the bond mid-point); a pair counts when |d - cutoff| <= tol, i.e. |d2 - cut2| <= 2 cutoff tol (+ tol^2).  One workgroup per (graph, set);
// the cross sets 2 and 3 hold the same pairs.  The 32-neighbour caps (by index, not by distance) have no ties.
__global__ void k_edge_ties(GraphArgs A, int* log, int stride, float tol) {
  const int g = blockIdx.x, k = blockIdx.y;
  const EdgeSet& S = A.set[k];
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  int c = 0;
  if (S.cap > 0) {
    const int l0 = A.b.lig_ptr[g], nl = A.b.lig_ptr[g + 1] - l0, a0 = A.b.atm_ptr[g], na = A.b.atm_ptr[g + 1] - a0;
    const float* LP = A.b.lig_pos;
    const float* RP = A.b.rec_pos;
    auto near = [&](float d2, float cut2, float t) { const float cut = sqrtf(cut2); return fabsf(d2 - cut2) <= 2.f * cut * t + t * t; };
    if (k == SET_LL || k == SET_AA) {
      const float* P = k == SET_LL ? LP : RP;
      const int p0 = k == SET_LL ? l0 : a0, np = k == SET_LL ? nl : na;
      const float cut2 = k == SET_LL ? A.lig_cut2 : A.atom_cut2;
      for (long idx = threadIdx.x; idx < (long)np * np; idx += blockDim.x) {
        const int i = (int)(idx / np), j = (int)(idx - (long)i * np);
        if (i < j) c += near(d2_rn(P[3 * (p0 + i)], P[3 * (p0 + i) + 1], P[3 * (p0 + i) + 2], P[3 * (p0 + j)], P[3 * (p0 + j) + 1], P[3 * (p0 + j) + 2]), cut2, tol);
      }
    } else if (k == SET_AL || k == SET_LA) {
      const float scale = A.dynamic_cross ? __fadd_rn(__fmul_rn(A.tr_sigma[g], 0.2f), 5.0f) : 1.0f;
      for (long idx = threadIdx.x; idx < (long)nl * na; idx += blockDim.x) {
        const int i = (int)(idx / na), j = (int)(idx - (long)i * na);
        if (A.is_cab[a0 + j]) continue;                        // (CA / CB atoms are edges whatever their distance)
        float x0 = LP[3 * (l0 + i)], y0 = LP[3 * (l0 + i) + 1], z0 = LP[3 * (l0 + i) + 2], x1 = RP[3 * (a0 + j)], y1 = RP[3 * (a0 + j) + 1], z1 = RP[3 * (a0 + j) + 2];
        if (A.dynamic_cross) { x0 = __fdiv_rn(x0, scale); y0 = __fdiv_rn(y0, scale); z0 = __fdiv_rn(z0, scale); x1 = __fdiv_rn(x1, scale); y1 = __fdiv_rn(y1, scale); z1 = __fdiv_rn(z1, scale); }
        c += near(d2_rn(x0, y0, z0, x1, y1, z1), A.cross_cut2, tol / scale);
      }
    } else {
      const bool tor = k == SET_TOR;
      const int t0 = tor ? A.b.tor_ptr[g] : A.b.sc_ptr[g], nt = (tor ? A.b.tor_ptr[g + 1] : A.b.sc_ptr[g + 1]) - t0;
      const float* P = tor ? LP : RP;
      const int p0 = tor ? l0 : a0, np = tor ? nl : na;
      const float cut2 = tor ? A.lig_cut2 : A.atom_cut2;
      for (long idx = threadIdx.x; idx < (long)nt * np; idx += blockDim.x) {
        const int t = (int)(idx / np), j = (int)(idx - (long)t * np);
        int u, v;
        if (tor) { const int b = A.b.tor_bond[t0 + t]; u = A.b.bond_src[b]; v = A.b.bond_dst[b]; }
        else { u = A.b.sc_bond[2 * (t0 + t)]; v = A.b.sc_bond[2 * (t0 + t) + 1]; }
        const float mx = (P[3 * u] + P[3 * v]) / 2, my = (P[3 * u + 1] + P[3 * v + 1]) / 2, mz = (P[3 * u + 2] + P[3 * v + 2]) / 2;
        c += near(d2_rn(mx, my, mz, P[3 * (p0 + j)], P[3 * (p0 + j) + 1], P[3 * (p0 + j) + 2]), cut2, tol);
      }
    }
  }
  if (c) atomicAdd(&total, c);
  __syncthreads();
  if (threadIdx.x == 0) log[k * stride + g] = total;
}

void launch_edge_ties(const GraphArgs& A, int* log_row, int stride, float tol, hipStream_t st) {
  hipLaunchKernelGGL(k_edge_ties, dim3(A.b.G, N_SETS), dim3(256), 0, st, A, log_row, stride, tol);
}

// ------------------------------------------------------------------------------------------------
// batch vectors / static per-batch tables
__global__ void k_batch_vectors(dbfr_batch b, int* lig_batch, int* atm_batch, uint8_t* is_cab, int* n_cab,
                                int* tor_batch, int* sc_batch) {
  const int g = blockIdx.x;
  for (int i = b.lig_ptr[g] + threadIdx.x; i < b.lig_ptr[g + 1]; i += blockDim.x) lig_batch[i] = g;
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int c = 0;
  for (int i = b.atm_ptr[g] + threadIdx.x; i < b.atm_ptr[g + 1]; i += blockDim.x) {
    atm_batch[i] = g;
    int a37 = (int)b.pocket_feat[5 * i];
    uint8_t f = (a37 == 1 || a37 == 3);  // protein_constants.atom_order['CA'], ['CB']
    is_cab[i] = f;
    c += f;
  }
  atomicAdd(&cnt, c);
  for (int i = b.tor_ptr[g] + threadIdx.x; i < b.tor_ptr[g + 1]; i += blockDim.x) tor_batch[i] = g;
  for (int i = b.sc_ptr[g] + threadIdx.x; i < b.sc_ptr[g + 1]; i += blockDim.x) sc_batch[i] = g;
  __syncthreads();
  if (threadIdx.x == 0) n_cab[g] = cnt;
}

void launch_batch_vectors(const dbfr_batch& b, int* lig_batch, int* atm_batch, uint8_t* is_cab, int* n_cab,
                          int* tor_batch, int* sc_batch, hipStream_t st) {
  hipLaunchKernelGGL(k_batch_vectors, dim3(b.G), dim3(128), 0, st, b, lig_batch, atm_batch, is_cab, n_cab, tor_batch,
                     sc_batch);
}

// sinusoidal_embedding(emb_scale * t, 32): time_emb.py:9-26 (evaluated in double, rounded once)
__global__ void k_time_embed(const float* t, int G, float emb_scale, float* temb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * EMB) return;
  int g = i / EMB, k = i % EMB;
  const int half = EMB / 2;
  int kk = k % half;
  float neg = -(float)(9.210340371976184 / (half - 1));        // -log(10000)/(half-1) as fp32
  float arg = (float)kk * neg;
  float f = (float)exp((double)arg);
  float x = (emb_scale * t[g]) * f;
  temb[i] = (float)(k < half ? sin((double)x) : cos((double)x));
}

void launch_time_embed(const float* t, int G, float emb_scale, float* temb, hipStream_t st) {
  hipLaunchKernelGGL(k_time_embed, dim3((G * EMB + 255) / 256), dim3(256), 0, st, t, G, emb_scale, temb);
}

// ------------------------------------------------------------------------------------------------
// Row-tiled 2-layer MLP (SimpleLinear, relu) with on-the-fly input assembly.
//   IN_LIGNODE : [lig_node(27) | temb[g]]                rows = ligand atoms
//   IN_LIGEDGE : [bond_feat(10) or 0 | temb[g] | gauss]  rows = edges of the ligand set
//   IN_TG      : [temb[g] | gauss]                       rows = edges (atom / cross / centre sets)
//   IN_G       : [gauss]                                 rows = edges (pseudotorque sets)
enum MlpIn { IN_LIGNODE = 0, IN_LIGEDGE = 1, IN_TG = 2, IN_G = 3 };

struct MlpArgs {
  Mlp2 w;
  int mode;
  const int* n_rows_dev;   // device row count (edges) or null
  int n_rows_max;
  const float* temb;       // [G][EMB]
  const int* row_graph_tab;  // batch vector indexed by tgt (edges) or by row (nodes)
  const int* tgt;          // per-edge target (graph lookup) or null
  const int* aux;          // bond index per edge (IN_LIGEDGE)
  const float* dist;       // per-edge distance
  const float* bond_feat;  // [EB][nfeat]
  int nfeat;
  const float* lig_node;   // [NL][nfeat_node]
  int nnode;
  const float* gs_offset;  // [EMB] GaussianSmearing buffers
  const float* gs_coeff;   // [1]
  float* out;              // [rows][NS]
};

#define MLP_ROWS 64
#define MLP_MAXIN 80

__global__ __launch_bounds__(256) void k_mlp(MlpArgs a) {
  __shared__ float xin[MLP_ROWS][MLP_MAXIN + 1];
  __shared__ float hid[MLP_ROWS][NS + 1];
  __shared__ float w0[MLP_MAXIN * NS];
  __shared__ float w1[NS * NS];
  __shared__ float s_gs[EMB];
  const int nrows = a.n_rows_dev ? min(*a.n_rows_dev, a.n_rows_max) : a.n_rows_max;
  if ((int)blockIdx.x * MLP_ROWS >= nrows) return;
  const int in = a.w.in;
  // weights once per workgroup, then a grid-stride walk over the 64-row tiles (one tile per workgroup cost 24 KB of weight staging per
  // 64 rows: 275 us per call on average at the bench batch, 2 % of the step)
  for (int i = threadIdx.x; i < in * NS; i += 256) w0[i] = a.w.w0t[i];
  for (int i = threadIdx.x; i < NS * NS; i += 256) w1[i] = a.w.w1t[i];
  const float coeff = a.gs_coeff ? a.gs_coeff[0] : 0.f;
  if (threadIdx.x < EMB) s_gs[threadIdx.x] = a.gs_offset ? a.gs_offset[threadIdx.x] : 0.f;
  __syncthreads();
  // zero padding of the reduction dim to a multiple of 4 (MFMA k-step)
  const int kin = (in + 3) & ~3;
  if (kin > in)
    for (int i = threadIdx.x; i < (kin - in) * NS; i += 256) w0[in * NS + i] = 0.f;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
  const int row = 16 * wave + n;
  for (int r0 = blockIdx.x * MLP_ROWS; r0 < nrows; r0 += gridDim.x * MLP_ROWS) {
  const int nr = min(MLP_ROWS, nrows - r0);
  {   // input assembly: four threads per row (columns q, q + 4, ...); what belongs to the row -- its graph, its clamped distance -- is
      // fetched once per thread, the Gaussian offsets come from LDS (element-wise assembly with a division and 3-4 dependent global
      // loads per element was most of this kernel's time)
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    const bool live = r < nr;
    const int arow = r0 + (live ? r : 0);
    const int gph = a.mode == IN_G ? 0 : a.row_graph_tab[a.tgt ? a.tgt[arow] : arow];
    const float dcl = a.dist ? fminf(a.dist[arow], s_gs[EMB - 1]) : 0.f;
    const int bond = a.mode == IN_LIGEDGE ? a.aux[arow] : -1;
    for (int c = q; c < in; c += 4) {
      float v = 0.f;
      if (live) {
        int cc = c;
        if (a.mode == IN_LIGNODE) {
          v = cc < a.nnode ? a.lig_node[(size_t)arow * a.nnode + cc] : a.temb[gph * EMB + cc - a.nnode];
        } else {
          if (a.mode == IN_LIGEDGE) {
            if (cc < a.nfeat) { v = bond >= 0 ? a.bond_feat[(size_t)bond * a.nfeat + cc] : 0.f; cc = -1; }
            else cc -= a.nfeat;
          }
          if (cc >= 0) {
            if (a.mode != IN_G && cc < EMB) v = a.temb[gph * EMB + cc];
            else {
              const int k = a.mode == IN_G ? cc : cc - EMB;
              const float d = dcl - s_gs[k];
              v = expf(coeff * (d * d));
            }
          }
        }
      }
      xin[r][c] = v;
    }
  }
  if (kin > in)
    for (int i = threadIdx.x; i < MLP_ROWS * (kin - in); i += 256) xin[i / (kin - in)][in + i % (kin - in)] = 0.f;
  __syncthreads();     // (also: every wave has finished reading `hid` of the previous tile)
  // both layers on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products): wave w owns rows 16w..16w+15,
  // D[channel, row] orientation => lane (g, n) holds 4 consecutive output channels of row 16w + n
  f32x4 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = a.w.b0 ? a.w.b0[16 * t + 4 * g + r] : 0.f;
  for (int k = 0; k < kin; k += 4) {
    const float bv = xin[row][k + g];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[(k + g) * NS + 16 * t + n], bv, acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) hid[row][16 * t + 4 * g + r] = fmaxf(acc[t][r], 0.f);
  __syncthreads();     // (also: every wave has finished reading `xin` of this tile)
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = a.w.b1 ? a.w.b1[16 * t + 4 * g + r] : 0.f;
#pragma unroll
  for (int k = 0; k < NS; k += 4) {
    const float bv = hid[row][k + g];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[(k + g) * NS + 16 * t + n], bv, acc[t], 0, 0, 0);
  }
  if (row < nr) {
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<f32x4*>(a.out + (size_t)(r0 + row) * NS + 16 * t + 4 * g) = acc[t];
  }
  }
}

void launch_mlp(const MlpArgs& a, hipStream_t st) {
  if (a.n_rows_max <= 0) return;
  const int tiles = (a.n_rows_max + MLP_ROWS - 1) / MLP_ROWS;
  hipLaunchKernelGGL(k_mlp, dim3(tiles < 1024 ? tiles : 1024), dim3(256), 0, st, a);    // 58 KB of LDS: two workgroups per CU, four tile sequences per CU
}

// AtomEncoder (equibind_encoder.py:68-88): sum of 5 categorical embeddings, then
// x += Linear_nobias([x | temb[g]])   (scalar_dim = 0 + sigma_embed_dim)
struct AtomEncArgs {
  const float* pocket_feat;  // [NA][5]
  const int* atm_batch;
  const float* temb;
  const float* emb[5];       // [dim_i][NS]
  int dims[5];
  const float* lin_t;        // [NS+EMB][NS] transposed scalar_lin.weight
  int NA;
  float* out;                // [NA][NS]
};

__global__ __launch_bounds__(256) void k_atom_encoder(AtomEncArgs a) {
  __shared__ float w[(NS + EMB) * NS];
  __shared__ float x[16][NS + EMB];
  for (int i = threadIdx.x; i < (NS + EMB) * NS; i += 256) w[i] = a.lin_t[i];
  const int r0 = blockIdx.x * 16;
  for (int i = threadIdx.x; i < 16 * (NS + EMB); i += 256) {
    int r = i / (NS + EMB), c = i % (NS + EMB);
    int row = r0 + r;
    float v = 0.f;
    if (row < a.NA) {
      if (c < NS) {
        for (int f = 0; f < 5; ++f) {
          int id = (int)a.pocket_feat[(size_t)row * 5 + f];
          id = min(max(id, 0), a.dims[f] - 1);
          v += a.emb[f][(size_t)id * NS + c];
        }
      } else {
        v = a.temb[a.atm_batch[row] * EMB + c - NS];
      }
    }
    x[r][c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * NS; i += 256) {
    int r = i / NS, c = i % NS;
    int row = r0 + r;
    if (row >= a.NA) continue;
    float acc = 0.f;
    for (int k = 0; k < NS + EMB; ++k) acc += w[k * NS + c] * x[r][k];
    a.out[(size_t)row * NS + c] = x[r][c] + acc;
  }
}

void launch_atom_encoder(const AtomEncArgs& a, hipStream_t st) {
  if (a.NA <= 0) return;
  hipLaunchKernelGGL(k_atom_encoder, dim3((a.NA + 15) / 16), dim3(256), 0, st, a);
}
