// MDN pose scorer for gfx950: the network forward of DiffBindFR's KarmaDock scorer
//   KarmaDock.forward            DiffBindFR/scoring/architecture/KarmaDock_sc.py:58-101
//   GraghTransformer (ligand)    GraphTransformer_Block.py:56-88,164-222,314-352,406-424   6 layers, 128 wide, 4 heads
//   GVP_embedding   (pocket)     GVP_Block.py:53-67,163-190,235-244,300-317,431-466        3 GVPConvLayers, (128, 16) dims
//   MDN_Block + score            MDN_Block.py:21-79, KarmaDock_sc.py:86-101                10 Gaussians per (ligand atom, residue) pair
// in eval mode.  Layout: graphs are contiguous node ranges (CSR pointers); pocket edges arrive grouped by target residue
// (knn_graph order), ligand bond edges in any order (a per-target incidence list is built once on the host side).
// Dense layers run on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products) through ONE gathered-concat GEMM
// kernel (k_lin: the concatenations [s_j | e | s_i | |V|] of the GVP message function are never materialised); BatchNorm
// (eval) is folded into the adjacent Linear at model creation; the pair feature W [h_l | h_t] of the mixture head is
// split into W_l h_l + W_t h_t (two small GEMMs) so that the per-pair work is 3 x (10 x 128) instead of 256 x 128.
// Every reduction has a fixed order (no atomics): scores are bitwise reproducible.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dbfr.h"

void dbfr_set_error(const std::string& s);
#define MCHECK(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) { dbfr_set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); return DBFR_ERR_HIP; } \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ generic dense layer
struct Seg { const float* p; int ld; int w; const int* idx; };   // row r reads p[(idx ? idx[r] : r) * ld + 0 .. w)
struct LinIn { Seg s[4]; int ns; };
struct Lin { const float* W; int ldw; const float* b; int N, K; };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

__device__ __forceinline__ float lin_in(const LinIn& in, int row, int k) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < in.ns) {
      if (k < in.s[i].w) return in.s[i].p[(size_t)(in.s[i].idx ? in.s[i].idx[row] : row) * in.s[i].ld + k];
      k -= in.s[i].w;
    }
  }
  return 0.f;
}

// out[m][n] = act(sum_k X[m][k] W[n][k] + b[n]) (+ res[m][n]); X = concatenation of up to 4 gathered segments.
// Workgroup = 4 waves = 64 rows x 64 output features; MFMA orientation D[feature, row]: a lane ends up with 4 consecutive
// features of one row (one 16-byte store).
__global__ __launch_bounds__(256) void k_lin(LinIn in, Lin w, int M, float* __restrict__ out, int ldo, int act,
                                             const float* __restrict__ res, int ldr) {
  __shared__ float Xs[64][17];
  __shared__ float Ws[64][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int n = lane & 15, g = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < w.K; k0 += 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * j, r = i >> 4, k = i & 15;
      const int row = m0 + r, col = n0 + r;
      Xs[r][k] = (row < M && k0 + k < w.K) ? lin_in(in, row, k0 + k) : 0.f;
      Ws[r][k] = (col < w.N && k0 + k < w.K) ? w.W[(size_t)col * w.ldw + k0 + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float b = Xs[16 * wave + n][4 * ks + g];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ws[16 * cb + n][4 * ks + g], b, acc[cb], 0, 0, 0);
    }
    __syncthreads();
  }
  const int row = m0 + 16 * wave + n;
  if (row >= M) return;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = n0 + 16 * cb + 4 * g + r;
      if (col >= w.N) continue;
      float v = acc[cb][r] + (w.b ? w.b[col] : 0.f);
      if (act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (act == ACT_SILU) v = v / (1.f + expf(-v));
      if (res) v = res[(size_t)row * ldr + col] + v;
      out[(size_t)row * ldo + col] = v;
    }
}

static void launch_lin(const LinIn& in, const Lin& w, int M, float* out, int ldo, int act, const float* res, int ldr, hipStream_t st) {
  if (M <= 0) return;
  hipLaunchKernelGGL(k_lin, dim3((M + 63) / 64, (w.N + 63) / 64), dim3(256), 0, st, in, w, M, out, ldo, act, res, ldr);
}
static LinIn in1(const float* p, int ld, int w, const int* idx = nullptr) { LinIn i; memset(&i, 0, sizeof i); i.ns = 1; i.s[0] = {p, ld, w, idx}; return i; }
static LinIn in_add(LinIn i, const float* p, int ld, int w, const int* idx = nullptr) { i.s[i.ns++] = {p, ld, w, idx}; return i; }

// ------------------------------------------------------------------------------------------------ graph transformer
// per (edge, head): alpha = clamp(K[row] * Q[col] / sqrt(d), +-5) * Eproj ; e_out = alpha ; ax = exp(clamp(sum_d alpha, +-5))
__global__ void k_gt_edge(const float* __restrict__ Q, const float* __restrict__ Kx, const float* __restrict__ Ep,
                          const int* __restrict__ row, const int* __restrict__ col, int E, float* __restrict__ e_out,
                          float* __restrict__ ax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E * 4) return;
  const int e = i >> 2, h = i & 3;
  const float* k = Kx + (size_t)row[e] * 128 + 32 * h;
  const float* q = Q + (size_t)col[e] * 128 + 32 * h;
  const float* p = Ep + (size_t)e * 128 + 32 * h;
  const float inv = 1.0f / sqrtf(32.f);
  float s = 0.f;
  for (int d = 0; d < 32; ++d) {
    float a = fminf(fmaxf(k[d] * q[d] * inv, -5.f), 5.f) * p[d];
    if (e_out) e_out[(size_t)e * 128 + 32 * h + d] = a;
    s += a;
  }
  ax[i] = expf(fminf(fmaxf(s, -5.f), 5.f));
}

// per (node, channel): h = sum_in V[row] ax / (sum_in ax + 1e-6), incoming edges in edge order (in_ptr / in_edge)
__global__ void k_gt_node(const float* __restrict__ V, const float* __restrict__ ax, const int* __restrict__ row,
                          const int* __restrict__ in_ptr, const int* __restrict__ in_edge, int N, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 128) return;
  const int node = i >> 7, c = i & 127, h = c >> 5;
  float wv = 0.f, z = 0.f;
  for (int k = in_ptr[node]; k < in_ptr[node + 1]; ++k) {
    const int e = in_edge[k];
    const float a = ax[4 * e + h];
    wv += V[(size_t)row[e] * 128 + c] * a;
    z += a;
  }
  out[i] = wv / (z + 1e-6f);
}

// ------------------------------------------------------------------------------------------------ GVP pieces
struct VSeg { const float* p; int nv; const int* idx; };    // row r reads p[(idx ? idx[r] : r) * nv * 3 ...]: nv vectors
struct GvpVec {
  VSeg s[3]; int ns;
  const float* wh; int vi, h;      // [h][vi]
  const float* wv; int vo;         // [vo][h] or null
  int gate;                        // v *= sigmoid(|v|)
  float* vn;                       // [M][h]   |V_h| per hidden vector channel (feeds the scalar Linear)
  float* vout;                     // [M][vo][3] or null
  int M;
};
#define GV_MAX 33

// thread per row: Vh = wh V (3 x h), vn = |Vh|, Vout = wv Vh (gated).  Weights in LDS.
__global__ __launch_bounds__(128) void k_gvp_vec(GvpVec a) {
  __shared__ float s_wh[GV_MAX * GV_MAX];
  __shared__ float s_wv[GV_MAX * GV_MAX];
  for (int i = threadIdx.x; i < a.h * a.vi; i += blockDim.x) s_wh[i] = a.wh[i];
  if (a.wv) for (int i = threadIdx.x; i < a.vo * a.h; i += blockDim.x) s_wv[i] = a.wv[i];
  __syncthreads();
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.M) return;
  float vh[GV_MAX][3];
  for (int j = 0; j < a.h; ++j) vh[j][0] = vh[j][1] = vh[j][2] = 0.f;
  int c0 = 0;
  for (int si = 0; si < a.ns; ++si) {
    const VSeg& S = a.s[si];
    const float* v = S.p + (size_t)(S.idx ? S.idx[r] : r) * S.nv * 3;
    for (int i = 0; i < S.nv; ++i) {
      const float x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
      for (int j = 0; j < a.h; ++j) {
        const float w = s_wh[j * a.vi + c0 + i];
        vh[j][0] += w * x; vh[j][1] += w * y; vh[j][2] += w * z;
      }
    }
    c0 += S.nv;
  }
  for (int j = 0; j < a.h; ++j)
    a.vn[(size_t)r * a.h + j] = sqrtf(fmaxf(vh[j][0] * vh[j][0] + vh[j][1] * vh[j][1] + vh[j][2] * vh[j][2], 1e-8f));
  if (!a.vout) return;
  for (int o = 0; o < a.vo; ++o) {
    float x = 0.f, y = 0.f, z = 0.f;
    for (int j = 0; j < a.h; ++j) { const float w = s_wv[o * a.h + j]; x += w * vh[j][0]; y += w * vh[j][1]; z += w * vh[j][2]; }
    if (a.gate) {
      const float nrm = sqrtf(fmaxf(x * x + y * y + z * z, 1e-8f));
      const float sg = 1.f / (1.f + expf(-nrm));
      x *= sg; y *= sg; z *= sg;
    }
    float* o3 = a.vout + ((size_t)r * a.vo + o) * 3;
    o3[0] = x; o3[1] = y; o3[2] = z;
  }
}

// LayerNorm for (s, V) tuples with an optional residual: s = LN(s_in + ds), V = (v_in + dv) / sqrt(mean_i max(|V_i|^2, eps)).
// One wave per row (ns <= 128).
__global__ __launch_bounds__(256) void k_gvp_ln(const float* __restrict__ s_in, const float* __restrict__ ds, int ns,
                                                const float* __restrict__ v_in, const float* __restrict__ dv, int nv,
                                                const float* __restrict__ lw, const float* __restrict__ lb, int M,
                                                float* __restrict__ s_out, float* __restrict__ v_out) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  float x[2];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = lane + 64 * j;
    x[j] = c < ns ? s_in[(size_t)r * ns + c] + (ds ? ds[(size_t)r * ns + c] : 0.f) : 0.f;
    sum += x[j];
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)ns;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int c = lane + 64 * j; if (c < ns) { const float d = x[j] - mean; sq += d * d; } }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float inv = 1.0f / sqrtf(sq / (float)ns + 1e-5f);
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int c = lane + 64 * j; if (c < ns) s_out[(size_t)r * ns + c] = (x[j] - mean) * inv * lw[c] + lb[c]; }
  if (!nv) return;
  float vx = 0.f, vy = 0.f, vz = 0.f, n2 = 0.f;
  if (lane < nv) {
    const float* p = v_in + ((size_t)r * nv + lane) * 3;
    vx = p[0]; vy = p[1]; vz = p[2];
    if (dv) { const float* q = dv + ((size_t)r * nv + lane) * 3; vx += q[0]; vy += q[1]; vz += q[2]; }
    n2 = fmaxf(vx * vx + vy * vy + vz * vz, 1e-8f);
  }
  for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o);
  const float vn = sqrtf(n2 / (float)nv);
  if (lane < nv) { float* o3 = v_out + ((size_t)r * nv + lane) * 3; o3[0] = vx / vn; o3[1] = vy / vn; o3[2] = vz / vn; }
}

// mean over the incoming edges of every node (edges grouped by target: ptr), channels = ns scalars | nv*3 vector comps
__global__ void k_mean_in(const float* __restrict__ ms, int ns, const float* __restrict__ mv, int nv3, const int* __restrict__ ptr,
                          int N, float* __restrict__ ds, float* __restrict__ dv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int C = ns + nv3;
  if (i >= N * C) return;
  const int node = i / C, c = i - node * C;
  const int b = ptr[node], e = ptr[node + 1];
  float acc = 0.f;
  if (c < ns) for (int k = b; k < e; ++k) acc += ms[(size_t)k * ns + c];
  else for (int k = b; k < e; ++k) acc += mv[(size_t)k * nv3 + c - ns];
  acc /= (float)max(e - b, 1);
  if (c < ns) ds[(size_t)node * ns + c] = acc; else dv[(size_t)node * nv3 + c - ns] = acc;
}

__global__ void k_seq_concat(const float* __restrict__ node_s, const int* __restrict__ seq, const float* __restrict__ Ws, int NR,
                             float* __restrict__ out) {   // [node_s(9) | W_s[seq](31)]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NR * 40) return;
  const int r = i / 40, c = i - r * 40;
  out[i] = c < 9 ? node_s[r * 9 + c] : Ws[min(max(seq[r], 0), 30) * 31 + c - 9];
}

// ------------------------------------------------------------------------------------------------ mixture-density head
// One workgroup per (graph, ligand atom); thread per residue: c = elu(Al[l] + At[t]) (BatchNorm already folded),
// z = Wz c (30 x 128, weights in LDS), density of the pair distance under the 10-component mixture, summed over the
// residues in a fixed order.  Distances as the reference computes them: float64 |x|^2 + |y|^2 - 2 x.y, sqrt, NaN -> 1e4,
// min over the 14 atom slots (unused slots are not masked).
__global__ __launch_bounds__(128) void k_mdn_pairs(const float* __restrict__ Al, const float* __restrict__ At,
                                                   const float* __restrict__ Wz, const float* __restrict__ bz,
                                                   const float* __restrict__ lig_pos, const float* __restrict__ xyz,
                                                   const int* __restrict__ lig_ptr, const int* __restrict__ res_ptr,
                                                   const int* __restrict__ lig_graph, float dist_thr, double* __restrict__ part) {
  __shared__ float s_w[30 * 128];
  __shared__ float s_al[128];
  __shared__ double s_sum[128];
  const int l = blockIdx.x, b = lig_graph[l];
  for (int i = threadIdx.x; i < 30 * 128; i += blockDim.x) s_w[i] = Wz[i];
  for (int i = threadIdx.x; i < 128; i += blockDim.x) s_al[i] = Al[(size_t)l * 128 + i];
  __syncthreads();
  const int r0 = res_ptr[b], nr = res_ptr[b + 1] - r0;
  const double lx = lig_pos[3 * l], ly = lig_pos[3 * l + 1], lz = lig_pos[3 * l + 2];
  const double l2 = lx * lx + ly * ly + lz * lz;
  double mine = 0.0;
  for (int t = threadIdx.x; t < nr; t += blockDim.x) {
    const float* at = At + (size_t)(r0 + t) * 128;
    float z[30];
#pragma unroll
    for (int j = 0; j < 30; ++j) z[j] = bz[j];
    for (int k = 0; k < 128; ++k) {
      float c = s_al[k] + at[k];
      c = c > 0.f ? c : expm1f(c);                       // ELU
#pragma unroll
      for (int j = 0; j < 30; ++j) z[j] += s_w[j * 128 + k] * c;
    }
    double dmin = 1e300;
    const float* p = xyz + (size_t)(r0 + t) * 42;
    for (int a = 0; a < 14; ++a) {
      const double x = p[3 * a], y = p[3 * a + 1], zz = p[3 * a + 2];
      const double d2 = -2.0 * (lx * x + ly * y + lz * zz) + (x * x + y * y + zz * zz) + l2;
      double d = sqrt(d2);
      if (d != d) d = 10000.0;
      dmin = fmin(dmin, d);
    }
    if (dmin > (double)dist_thr) continue;
    float mx = z[0];
#pragma unroll
    for (int j = 1; j < 10; ++j) mx = fmaxf(mx, z[j]);
    float den = 0.f, pe[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) { pe[j] = expf(z[j] - mx); den += pe[j]; }
    double prob = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const float pi = pe[j] / den;
      const float sg = (z[10 + j] > 0.f ? z[10 + j] : expm1f(z[10 + j])) + 1.1f;
      const float mu = (z[20 + j] > 0.f ? z[20 + j] : expm1f(z[20 + j])) + 1.0f;
      const float var2 = 2.f * (sg * sg);
      const double dd = dmin - (double)mu;
      const double logp = -(dd * dd) / (double)var2 - (double)logf(sg) - 0.9189385332046727 + (double)logf(pi);
      prob += exp(logp);
    }
    mine += prob;
  }
  s_sum[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < 128; ++i) tot += s_sum[i];
    part[l] = tot;
  }
}

__global__ void k_mdn_sum(const double* __restrict__ part, const int* __restrict__ lig_ptr, int B, float* __restrict__ score) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double tot = 0.0;
  for (int l = lig_ptr[b]; l < lig_ptr[b + 1]; ++l) tot += part[l];
  score[b] = (float)tot;
}

__global__ void k_graph_of(const int* __restrict__ ptr, int B, int* __restrict__ graph) {
  const int b = blockIdx.x;
  for (int i = ptr[b] + threadIdx.x; i < ptr[b + 1]; i += blockDim.x) graph[i] = b;
}

// ------------------------------------------------------------------------------------------------ model (host side)
struct GvpW { Lin ws; const float* wh; const float* wv; int si, vi, so, vo, h; };
struct LnW { const float* w; const float* b; };
struct GtLayer { Lin q, k, v, ep, o_node, mlp0, mlp3, o_edge, emlp0, emlp3; };
struct GvpLayer { GvpW m0, m1, m2, f0, f1; LnW n0, n1; };

struct dbfr_mdn_model {
  std::vector<void*> allocs;
  Lin node_enc, edge_enc;
  GtLayer gt[6];
  const float* W_s;
  LnW ln_v, ln_e, ln_out;
  GvpW wv1, we1, wout;
  GvpLayer layer[3];
  Lin mdn_l, mdn_t;
  const float *Wz, *bz;
};

typedef std::map<std::string, const dbfr_tensor*> TMapM;

namespace {
struct Packer {
  dbfr_mdn_model* m; const TMapM& tm; int rc = 0;
  const float* get(const std::string& n, int64_t numel) {
    auto it = tm.find(n);
    if (it == tm.end()) { if (!rc) { rc = DBFR_ERR_ARG; dbfr_set_error("missing tensor '" + n + "'"); } return nullptr; }
    if (it->second->numel != numel) {
      if (!rc) { rc = DBFR_ERR_ARG; dbfr_set_error("tensor '" + n + "' has " + std::to_string(it->second->numel) + " elements, expected " + std::to_string(numel)); }
      return nullptr;
    }
    return it->second->data;
  }
  const float* up(const std::vector<float>& h) {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size() * 4, 16)) != hipSuccess) { if (!rc) { rc = DBFR_ERR_HIP; dbfr_set_error("hipMalloc failed (mdn weights)"); } return nullptr; }
    m->allocs.push_back(d);
    if (!h.empty() && hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { if (!rc) { rc = DBFR_ERR_HIP; dbfr_set_error("hipMemcpy failed (mdn weights)"); } }
    return (const float*)d;
  }
  const float* raw(const std::string& n, int64_t numel) { const float* p = get(n, numel); return p ? up(std::vector<float>(p, p + numel)) : nullptr; }
  // Linear, optionally with an eval-mode BatchNorm1d folded in FRONT (y = W bn(x) + b) or BEHIND (y = bn(W x + b))
  Lin lin(const std::string& n, int N, int K, bool bias, const std::string& bn_front = "", const std::string& bn_back = "") {
    Lin L; memset(&L, 0, sizeof L); L.N = N; L.K = K; L.ldw = K;
    const float* W = get(n + ".weight", (int64_t)N * K);
    const float* b = bias ? get(n + ".bias", N) : nullptr;
    if (!W || (bias && !b)) return L;
    std::vector<float> w(W, W + (size_t)N * K), bb(N, 0.f);
    if (b) for (int i = 0; i < N; ++i) bb[i] = b[i];
    bool has_b = bias;
    auto bn = [&](const std::string& k, int n_, std::vector<double>& a, std::vector<double>& c) {
      const float *g = get(k + ".weight", n_), *be = get(k + ".bias", n_), *mu = get(k + ".running_mean", n_), *var = get(k + ".running_var", n_);
      if (!g || !be || !mu || !var) return false;
      a.resize(n_); c.resize(n_);
      for (int i = 0; i < n_; ++i) { a[i] = (double)g[i] / std::sqrt((double)var[i] + 1e-5); c[i] = (double)be[i] - (double)mu[i] * a[i]; }
      return true;
    };
    if (!bn_front.empty()) {
      std::vector<double> a, c;
      if (!bn(bn_front, K, a, c)) return L;
      for (int i = 0; i < N; ++i) {
        double acc = bb[i];
        for (int k = 0; k < K; ++k) { acc += (double)w[(size_t)i * K + k] * c[k]; w[(size_t)i * K + k] = (float)((double)w[(size_t)i * K + k] * a[k]); }
        bb[i] = (float)acc;
      }
      has_b = true;
    }
    if (!bn_back.empty()) {
      std::vector<double> a, c;
      if (!bn(bn_back, N, a, c)) return L;
      for (int i = 0; i < N; ++i) {
        for (int k = 0; k < K; ++k) w[(size_t)i * K + k] = (float)((double)w[(size_t)i * K + k] * a[i]);
        bb[i] = (float)((double)bb[i] * a[i] + c[i]);
      }
      has_b = true;
    }
    L.W = up(w);
    L.b = has_b ? up(bb) : nullptr;
    return L;
  }
  GvpW gvp(const std::string& n, int si, int vi, int so, int vo) {
    GvpW G; memset(&G, 0, sizeof G);
    G.si = si; G.vi = vi; G.so = so; G.vo = vo; G.h = std::max(vi, vo);
    G.wh = raw(n + ".wh.weight", (int64_t)G.h * vi);
    G.ws = lin(n + ".ws", so, G.h + si, true);
    G.wv = vo ? raw(n + ".wv.weight", (int64_t)vo * G.h) : nullptr;
    return G;
  }
  LnW ln(const std::string& n, int w) { LnW L; L.w = raw(n + ".scalar_norm.weight", w); L.b = raw(n + ".scalar_norm.bias", w); return L; }
};
}  // namespace

extern "C" void dbfr_mdn_model_destroy(dbfr_mdn_model* m) {
  if (!m) return;
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
}

extern "C" int dbfr_mdn_model_create(const dbfr_tensor* tensors, int32_t n_tensors, dbfr_mdn_model** out) {
  try {
    if (!tensors || !out) { dbfr_set_error("null argument"); return DBFR_ERR_ARG; }
    TMapM tm;
    for (int i = 0; i < n_tensors; ++i) tm[tensors[i].name] = &tensors[i];
    dbfr_mdn_model* m = new dbfr_mdn_model();
    Packer P{m, tm};
    m->node_enc = P.lin("lig_encoder.node_encoder", 128, 89, true);
    m->edge_enc = P.lin("lig_encoder.edge_encoder", 128, 20, true);
    for (int l = 0; l < 6; ++l) {
      const std::string k = "lig_encoder.gt_block." + std::to_string(l);
      GtLayer& G = m->gt[l];
      const bool fin = l == 5;
      G.q = P.lin(k + ".mha_module.Q", 128, 128, false, k + ".batch_norm1_node_feats");
      G.k = P.lin(k + ".mha_module.K", 128, 128, false, k + ".batch_norm1_node_feats");
      G.v = P.lin(k + ".mha_module.V", 128, 128, false, k + ".batch_norm1_node_feats");
      G.ep = P.lin(k + ".mha_module.edge_feats_projection", 128, 128, false, k + ".batch_norm1_edge_feats");
      G.o_node = P.lin(k + ".O_node_feats", 128, 128, true);
      G.mlp0 = P.lin(k + ".node_feats_MLP.0", 256, 128, false, k + ".batch_norm2_node_feats");
      G.mlp3 = P.lin(k + ".node_feats_MLP.3", 128, 256, false);
      if (!fin) {
        G.o_edge = P.lin(k + ".O_edge_feats", 128, 128, true);
        G.emlp0 = P.lin(k + ".edge_feats_MLP.0", 256, 128, false, k + ".batch_norm2_edge_feats");
        G.emlp3 = P.lin(k + ".edge_feats_MLP.3", 128, 256, false);
      }
    }
    m->W_s = P.raw("pro_encoder.W_s.weight", 31 * 31);
    m->ln_v = P.ln("pro_encoder.W_v.0", 40); m->wv1 = P.gvp("pro_encoder.W_v.1", 40, 3, 128, 16);
    m->ln_e = P.ln("pro_encoder.W_e.0", 21); m->we1 = P.gvp("pro_encoder.W_e.1", 21, 1, 32, 1);
    for (int l = 0; l < 3; ++l) {
      const std::string k = "pro_encoder.layers." + std::to_string(l);
      GvpLayer& G = m->layer[l];
      G.m0 = P.gvp(k + ".conv.message_func.0", 288, 33, 128, 16);
      G.m1 = P.gvp(k + ".conv.message_func.1", 128, 16, 128, 16);
      G.m2 = P.gvp(k + ".conv.message_func.2", 128, 16, 128, 16);
      G.n0 = P.ln(k + ".norm.0", 128); G.n1 = P.ln(k + ".norm.1", 128);
      G.f0 = P.gvp(k + ".ff_func.0", 128, 16, 512, 32);
      G.f1 = P.gvp(k + ".ff_func.1", 512, 32, 128, 16);
    }
    m->ln_out = P.ln("pro_encoder.W_out.0", 128); m->wout = P.gvp("pro_encoder.W_out.1", 128, 16, 128, 0);
    // pair MLP with its BatchNorm folded, split into the ligand half (no bias) and the residue half (+ bias)
    Lin full = P.lin("mdn_layer.MLP.0", 128, 256, true, "", "mdn_layer.MLP.1");
    m->mdn_l = full; m->mdn_l.K = 128; m->mdn_l.b = nullptr;
    m->mdn_t = full; m->mdn_t.K = 128; m->mdn_t.W = full.W ? full.W + 128 : nullptr;
    {
      std::vector<float> wz(30 * 128), bz(30);
      const char* hn[3] = {"mdn_layer.z_pi", "mdn_layer.z_sigma", "mdn_layer.z_mu"};
      for (int h = 0; h < 3; ++h) {
        const float* W = P.get(std::string(hn[h]) + ".weight", 10 * 128);
        const float* b = P.get(std::string(hn[h]) + ".bias", 10);
        if (W && b) { memcpy(&wz[h * 1280], W, 1280 * 4); memcpy(&bz[h * 10], b, 40); }
      }
      m->Wz = P.up(wz); m->bz = P.up(bz);
    }
    if (P.rc) { dbfr_mdn_model_destroy(m); return P.rc; }
    *out = m;
    return DBFR_OK;
  } catch (const std::exception& e) {
    dbfr_set_error(std::string("dbfr_mdn_model_create: ") + e.what());
    return DBFR_ERR_ARG;
  }
}

// ------------------------------------------------------------------------------------------------ forward
struct MBump {
  char* base; size_t off;
  float* f(size_t n) { off = (off + 255) & ~(size_t)255; float* p = base ? (float*)(base + off) : nullptr; off += n * 4; return p; }
  int* i(size_t n) { return (int*)f(n); }
  double* d(size_t n) { return (double*)f(2 * n); }
};

struct MWs {
  float *x[2], *e[2], *Q, *K, *V, *Ep, *eatt, *ax, *hatt, *t256, *te256;
  float *s40, *s[2], *v[2], *vn_n, *es0, *ev0, *es, *ev, *vn_e, *ms[2], *mv[2], *ds, *dv, *fs, *fv, *vn_f, *pro_s, *lig_s;
  float *Al, *At; double* part; int* lig_graph;
};

static size_t mdn_plan(const dbfr_mdn_batch* B, char* base, MWs* w) {
  MBump b{base, 0};
  const size_t NL = B->NL, EL = std::max(B->EL, 1), NR = B->NR, EP = std::max(B->EP, 1);
  for (int i = 0; i < 2; ++i) { w->x[i] = b.f(NL * 128); w->e[i] = b.f(EL * 128); }
  w->Q = b.f(NL * 128); w->K = b.f(NL * 128); w->V = b.f(NL * 128); w->Ep = b.f(EL * 128); w->eatt = b.f(EL * 128);
  w->ax = b.f(EL * 4); w->hatt = b.f(NL * 128); w->t256 = b.f(NL * 256); w->te256 = b.f(EL * 256);
  w->s40 = b.f(NR * 40);
  for (int i = 0; i < 2; ++i) { w->s[i] = b.f(NR * 128); w->v[i] = b.f(NR * 48); }
  w->vn_n = b.f(NR * 33); w->es0 = b.f(EP * 21); w->ev0 = b.f(EP * 3); w->es = b.f(EP * 32); w->ev = b.f(EP * 3); w->vn_e = b.f(EP * 33);
  for (int i = 0; i < 2; ++i) { w->ms[i] = b.f(EP * 128); w->mv[i] = b.f(EP * 48); }
  w->ds = b.f(NR * 128); w->dv = b.f(NR * 48); w->fs = b.f(NR * 512); w->fv = b.f(NR * 96); w->vn_f = b.f(NR * 33);
  w->pro_s = b.f(NR * 128); w->lig_s = b.f(NL * 128);
  w->Al = b.f(NL * 128); w->At = b.f(NR * 128); w->part = b.d(NL); w->lig_graph = b.i(NL);
  return b.off + 256;
}

extern "C" int dbfr_mdn_workspace_bytes(const dbfr_mdn_batch* b, size_t* bytes) {
  if (!b || !bytes || b->B <= 0 || b->NL <= 0 || b->NR <= 0) { dbfr_set_error("bad mdn batch"); return DBFR_ERR_ARG; }
  MWs w;
  *bytes = mdn_plan(b, nullptr, &w);
  return DBFR_OK;
}

static void gvp_run(const GvpW& G, const GvpVec& vec_in, const LinIn& s_in, int M, int act_s, int gate, float* vn, float* s_out,
                    float* v_out, hipStream_t st) {
  GvpVec a = vec_in;
  a.wh = G.wh; a.vi = G.vi; a.h = G.h; a.wv = G.wv; a.vo = G.vo; a.gate = gate; a.vn = vn; a.vout = G.vo ? v_out : nullptr; a.M = M;
  if (M > 0) hipLaunchKernelGGL(k_gvp_vec, dim3((M + 127) / 128), dim3(128), 0, st, a);
  launch_lin(in_add(s_in, vn, G.h, G.h), G.ws, M, s_out, G.so, act_s, nullptr, 0, st);
}
static GvpVec vec1(const float* p, int nv, const int* idx = nullptr) { GvpVec a; memset(&a, 0, sizeof a); a.ns = 1; a.s[0] = {p, nv, idx}; return a; }

extern "C" int dbfr_mdn_forward(dbfr_mdn_model* m, const dbfr_mdn_batch* B, float* score, float* lig_s_out, float* pro_s_out,
                                void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!m || !B || !score || !workspace) { dbfr_set_error("null argument"); return DBFR_ERR_ARG; }
  if (B->B <= 0 || B->NL <= 0 || B->NR <= 0 || B->EL < 0 || B->EP < 0) { dbfr_set_error("bad mdn batch"); return DBFR_ERR_ARG; }
  MWs w;
  if (mdn_plan(B, (char*)workspace, &w) > workspace_bytes) { dbfr_set_error("mdn workspace too small"); return DBFR_ERR_ARG; }
  hipStream_t st = (hipStream_t)hip_stream;
  const int NL = B->NL, EL = B->EL, NR = B->NR, EP = B->EP;
  // ---------------- ligand encoder (pose independent: a caller scoring many poses of one ligand may pass lig_s_in)
  const float* lig_s = B->lig_s_in;
  if (!lig_s) {
    launch_lin(in1(B->lig_node_s, 89, 89), m->node_enc, NL, w.x[0], 128, ACT_NONE, nullptr, 0, st);
    launch_lin(in1(B->lig_edge_s, 20, 20), m->edge_enc, EL, w.e[0], 128, ACT_NONE, nullptr, 0, st);
    int cur = 0;
    for (int l = 0; l < 6; ++l) {
      const GtLayer& G = m->gt[l];
      const bool fin = l == 5;
      float *x = w.x[cur], *xn = w.x[cur ^ 1], *e = w.e[cur], *en = w.e[cur ^ 1];
      launch_lin(in1(x, 128, 128), G.q, NL, w.Q, 128, ACT_NONE, nullptr, 0, st);
      launch_lin(in1(x, 128, 128), G.k, NL, w.K, 128, ACT_NONE, nullptr, 0, st);
      launch_lin(in1(x, 128, 128), G.v, NL, w.V, 128, ACT_NONE, nullptr, 0, st);
      launch_lin(in1(e, 128, 128), G.ep, EL, w.Ep, 128, ACT_NONE, nullptr, 0, st);
      if (EL > 0) hipLaunchKernelGGL(k_gt_edge, dim3((EL * 4 + 255) / 256), dim3(256), 0, st, w.Q, w.K, w.Ep, B->lig_edge_src, B->lig_edge_dst, EL, fin ? nullptr : w.eatt, w.ax);
      hipLaunchKernelGGL(k_gt_node, dim3((NL * 128 + 255) / 256), dim3(256), 0, st, w.V, w.ax, B->lig_edge_src, B->lig_in_ptr, B->lig_in_edge, NL, w.hatt);
      launch_lin(in1(w.hatt, 128, 128), G.o_node, NL, xn, 128, ACT_NONE, x, 128, st);             // x1 = x + O(h)
      launch_lin(in1(xn, 128, 128), G.mlp0, NL, w.t256, 256, ACT_SILU, nullptr, 0, st);
      launch_lin(in1(w.t256, 256, 256), G.mlp3, NL, x, 128, ACT_NONE, xn, 128, st);               // x2 = x1 + MLP(bn2(x1)) -> back into x
      if (!fin) {
        launch_lin(in1(w.eatt, 128, 128), G.o_edge, EL, en, 128, ACT_NONE, e, 128, st);
        launch_lin(in1(en, 128, 128), G.emlp0, EL, w.te256, 256, ACT_SILU, nullptr, 0, st);
        launch_lin(in1(w.te256, 256, 256), G.emlp3, EL, e, 128, ACT_NONE, en, 128, st);
      }
      (void)cur;                                                                                  // results stay in slot `cur`
    }
    MCHECK(hipMemcpyAsync(w.lig_s, w.x[0], (size_t)NL * 128 * 4, hipMemcpyDeviceToDevice, st));
    lig_s = w.lig_s;
  }
  if (lig_s_out) MCHECK(hipMemcpyAsync(lig_s_out, lig_s, (size_t)NL * 128 * 4, hipMemcpyDeviceToDevice, st));
  // ---------------- pocket encoder
  hipLaunchKernelGGL(k_seq_concat, dim3((NR * 40 + 255) / 256), dim3(256), 0, st, B->pro_node_s, B->pro_seq, m->W_s, NR, w.s40);
  hipLaunchKernelGGL(k_gvp_ln, dim3((NR + 3) / 4), dim3(256), 0, st, w.s40, (const float*)nullptr, 40, B->pro_node_v, (const float*)nullptr, 3, m->ln_v.w, m->ln_v.b, NR, w.s40, w.v[1]);
  gvp_run(m->wv1, vec1(w.v[1], 3), in1(w.s40, 40, 40), NR, ACT_NONE, 0, w.vn_n, w.s[0], w.v[0], st);
  if (EP > 0) {
    hipLaunchKernelGGL(k_gvp_ln, dim3((EP + 3) / 4), dim3(256), 0, st, B->pro_edge_s, (const float*)nullptr, 21, B->pro_edge_v, (const float*)nullptr, 1, m->ln_e.w, m->ln_e.b, EP, w.es0, w.ev0);
    gvp_run(m->we1, vec1(w.ev0, 1), in1(w.es0, 21, 21), EP, ACT_NONE, 0, w.vn_e, w.es, w.ev, st);
  }
  const int *src = B->pro_edge_src, *dst = B->pro_edge_dst;
  int cur = 0;
  for (int l = 0; l < 3; ++l) {
    const GvpLayer& G = m->layer[l];
    float *s = w.s[cur], *v = w.v[cur], *sn = w.s[cur ^ 1], *vn = w.v[cur ^ 1];
    if (EP > 0) {
      GvpVec mv; memset(&mv, 0, sizeof mv); mv.ns = 3; mv.s[0] = {v, 16, src}; mv.s[1] = {w.ev, 1, nullptr}; mv.s[2] = {v, 16, dst};
      LinIn ms = in_add(in_add(in1(s, 128, 128, src), w.es, 32, 32), s, 128, 128, dst);
      gvp_run(G.m0, mv, ms, EP, ACT_RELU, 1, w.vn_e, w.ms[0], w.mv[0], st);
      gvp_run(G.m1, vec1(w.mv[0], 16), in1(w.ms[0], 128, 128), EP, ACT_RELU, 1, w.vn_e, w.ms[1], w.mv[1], st);
      gvp_run(G.m2, vec1(w.mv[1], 16), in1(w.ms[1], 128, 128), EP, ACT_NONE, 0, w.vn_e, w.ms[0], w.mv[0], st);
    }
    hipLaunchKernelGGL(k_mean_in, dim3((NR * 176 + 255) / 256), dim3(256), 0, st, w.ms[0], 128, w.mv[0], 48, B->pro_in_ptr, NR, w.ds, w.dv);
    hipLaunchKernelGGL(k_gvp_ln, dim3((NR + 3) / 4), dim3(256), 0, st, s, w.ds, 128, v, w.dv, 16, G.n0.w, G.n0.b, NR, sn, vn);
    gvp_run(G.f0, vec1(vn, 16), in1(sn, 128, 128), NR, ACT_RELU, 1, w.vn_f, w.fs, w.fv, st);
    gvp_run(G.f1, vec1(w.fv, 32), in1(w.fs, 512, 512), NR, ACT_NONE, 0, w.vn_f, w.ds, w.dv, st);
    hipLaunchKernelGGL(k_gvp_ln, dim3((NR + 3) / 4), dim3(256), 0, st, sn, w.ds, 128, vn, w.dv, 16, G.n1.w, G.n1.b, NR, s, v);
    (void)cur;                                                                                    // layer output back in slot `cur`
  }
  hipLaunchKernelGGL(k_gvp_ln, dim3((NR + 3) / 4), dim3(256), 0, st, w.s[0], (const float*)nullptr, 128, w.v[0], (const float*)nullptr, 16, m->ln_out.w, m->ln_out.b, NR, w.s[1], w.v[1]);
  gvp_run(m->wout, vec1(w.v[1], 16), in1(w.s[1], 128, 128), NR, ACT_RELU, 0, w.vn_n, w.pro_s, nullptr, st);
  if (pro_s_out) MCHECK(hipMemcpyAsync(pro_s_out, w.pro_s, (size_t)NR * 128 * 4, hipMemcpyDeviceToDevice, st));
  // ---------------- mixture-density head
  launch_lin(in1(lig_s, 128, 128), m->mdn_l, NL, w.Al, 128, ACT_NONE, nullptr, 0, st);
  launch_lin(in1(w.pro_s, 128, 128), m->mdn_t, NR, w.At, 128, ACT_NONE, nullptr, 0, st);
  hipLaunchKernelGGL(k_graph_of, dim3(B->B), dim3(64), 0, st, B->lig_ptr, B->B, w.lig_graph);
  hipLaunchKernelGGL(k_mdn_pairs, dim3(NL), dim3(128), 0, st, w.Al, w.At, m->Wz, m->bz, B->lig_pos, B->pro_xyz_full, B->lig_ptr, B->res_ptr, w.lig_graph, B->dist_threshold > 0 ? B->dist_threshold : 5.0f, w.part);
  hipLaunchKernelGGL(k_mdn_sum, dim3((B->B + 63) / 64), dim3(64), 0, st, w.part, B->lig_ptr, B->B, score);
  MCHECK(hipGetLastError());
  return DBFR_OK;
}


// ------------------------------------------------------------------------------------------------ pocket featurisation
// The pocket half of the scorer's input from residue types + atom14 coordinates (absent atoms at the origin), for any number
// of pockets / poses at once: `get_protein_feature` of DiffBindFR/scoring/dataset/protein_feature.py:137-216 behind its PDB
// parser (intra-residue distances, backbone dihedral sin/cos as openfold's atom37_to_torsion_angles computes them, CA
// orientations + side-chain direction, the k nearest CA neighbours of every residue, 21 edge scalars, unit CA-CA vector).
// Quirks kept as listed in oracle/mdn_features.py.  One workgroup per pocket, thread per residue.
#define PF_MAXR 1024
#define PF_MAXK 32
__constant__ int kIdealAtoms[21] = {5, 11, 8, 8, 6, 9, 9, 4, 10, 8, 8, 9, 8, 11, 7, 6, 7, 14, 12, 7, 0};

struct PocketFeatArgs {
  const int* res_ptr; const int* edge_ptr; const int* aatype; const float* x14; int topk; int n_res;
  float* node_s; float* node_v; int* edge_src; int* edge_dst; int* in_ptr; float* edge_s; float* edge_v;
};

__device__ __forceinline__ void dihedral_sc(const float* a0, const float* a1, const float* a2, const float* a3, float sign, float* out) {
  float e0[3], e1[3], e2[3], d[3];
  for (int c = 0; c < 3; ++c) { e0[c] = a2[c] - a1[c]; e1[c] = a0[c] - a2[c]; d[c] = a3[c] - a2[c]; }
  float den = sqrtf((e0[0] * e0[0] + e0[1] * e0[1]) + e0[2] * e0[2] + 1e-8f);
  for (int c = 0; c < 3; ++c) e0[c] = e0[c] / den;
  const float dot = (e0[0] * e1[0] + e0[1] * e1[1]) + e0[2] * e1[2];
  for (int c = 0; c < 3; ++c) e1[c] = e1[c] - e0[c] * dot;
  den = sqrtf((e1[0] * e1[0] + e1[1] * e1[1]) + e1[2] * e1[2] + 1e-8f);
  for (int c = 0; c < 3; ++c) e1[c] = e1[c] / den;
  e2[0] = e0[1] * e1[2] - e0[2] * e1[1]; e2[1] = e0[2] * e1[0] - e0[0] * e1[2]; e2[2] = e0[0] * e1[1] - e0[1] * e1[0];
  const float z = (e2[0] * d[0] + e2[1] * d[1]) + e2[2] * d[2], y = (e1[0] * d[0] + e1[1] * d[1]) + e1[2] * d[2];
  const float nn = sqrtf((z * z + y * y) + 1e-8f);
  out[0] = sign * (z / nn); out[1] = sign * (y / nn);
}
__device__ __forceinline__ void unit3(float x, float y, float z, float* o) {   // nan_to_num(v / |v|)
  const float n = sqrtf((x * x + y * y) + z * z);
  o[0] = n > 0.f ? x / n : 0.f; o[1] = n > 0.f ? y / n : 0.f; o[2] = n > 0.f ? z / n : 0.f;
}
__device__ __forceinline__ float norm_eps(float x, float y, float z) {          // |v + 1e-6|
  x += 1e-6f; y += 1e-6f; z += 1e-6f;
  return sqrtf((x * x + y * y) + z * z);
}

__global__ __launch_bounds__(128) void k_pocket_features(PocketFeatArgs a) {
  __shared__ float ca[PF_MAXR][3], cb[PF_MAXR][3], com[PF_MAXR][3];
  const int g = blockIdx.x, r0 = a.res_ptr[g], n = min(a.res_ptr[g + 1] - r0, PF_MAXR);
  const int k_eff = min(a.topk, n - 1);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float* X = a.x14 + (size_t)(r0 + i) * 42;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int s = 0; s < 14; ++s) { sx += X[3 * s]; sy += X[3 * s + 1]; sz += X[3 * s + 2]; }
    const float cnt = (float)kIdealAtoms[min(max(a.aatype[r0 + i], 0), 20)];
    com[i][0] = sx / cnt; com[i][1] = sy / cnt; com[i][2] = sz / cnt;
    for (int c = 0; c < 3; ++c) { ca[i][c] = X[3 + c]; cb[i][c] = X[12 + c]; }
  }
  __syncthreads();
  const float zero3[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int gi = r0 + i;
    const float* X = a.x14 + (size_t)gi * 42;
    const float *N_ = X, *CA = X + 3, *C = X + 6, *O = X + 9;
    const float* P = i > 0 ? X - 42 : nullptr;             // previous residue in array order (zeros before the first)
    float* ns = a.node_s + (size_t)gi * 9;
    ns[0] = 0.1f * norm_eps(CA[0] - O[0], CA[1] - O[1], CA[2] - O[2]);
    ns[1] = 0.1f * norm_eps(N_[0] - O[0], N_[1] - O[1], N_[2] - O[2]);
    ns[2] = 0.1f * norm_eps(N_[0] - C[0], N_[1] - C[1], N_[2] - C[2]);
    dihedral_sc(P ? P + 3 : zero3, P ? P + 6 : zero3, N_, CA, 1.f, ns + 3);     // pre-omega: CA-1, C-1, N, CA
    dihedral_sc(P ? P + 6 : zero3, N_, CA, C, 1.f, ns + 5);                     // phi: C-1, N, CA, C
    dihedral_sc(N_, CA, C, O, -1.f, ns + 7);                                    // psi: N, CA, C, O (negated)
    float* nv = a.node_v + (size_t)gi * 9;
    if (i + 1 < n) unit3(ca[i + 1][0] - ca[i][0], ca[i + 1][1] - ca[i][1], ca[i + 1][2] - ca[i][2], nv); else nv[0] = nv[1] = nv[2] = 0.f;
    if (i > 0) unit3(ca[i - 1][0] - ca[i][0], ca[i - 1][1] - ca[i][1], ca[i - 1][2] - ca[i][2], nv + 3); else nv[3] = nv[4] = nv[5] = 0.f;
    {
      float c[3], nn[3], bis[3], perp[3];
      unit3(C[0] - CA[0], C[1] - CA[1], C[2] - CA[2], c);
      unit3(N_[0] - CA[0], N_[1] - CA[1], N_[2] - CA[2], nn);
      unit3(c[0] + nn[0], c[1] + nn[1], c[2] + nn[2], bis);
      unit3(c[1] * nn[2] - c[2] * nn[1], c[2] * nn[0] - c[0] * nn[2], c[0] * nn[1] - c[1] * nn[0], perp);
      const float s13 = 0.5773502691896257f, s23 = 0.816496580927726f;
      for (int q = 0; q < 3; ++q) nv[6 + q] = -bis[q] * s13 - perp[q] * s23;
    }
    // k nearest other residues by CA distance (stable: ties keep the lower index first)
    float bd[PF_MAXK]; int bj[PF_MAXK]; int cnt = 0;
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;
      const float dx = ca[i][0] - ca[j][0], dy = ca[i][1] - ca[j][1], dz = ca[i][2] - ca[j][2];
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      if (cnt == k_eff && !(d2 < bd[cnt - 1])) continue;
      int p = cnt < k_eff ? cnt++ : cnt - 1;
      while (p > 0 && bd[p - 1] > d2) { bd[p] = bd[p - 1]; bj[p] = bj[p - 1]; --p; }
      bd[p] = d2; bj[p] = j;
    }
    const int e0 = a.edge_ptr[g] + i * k_eff;
    a.in_ptr[gi] = e0;
    for (int r = 0; r < k_eff; ++r) {
      const int j = bj[r], e = e0 + r;
      a.edge_src[e] = r0 + j; a.edge_dst[e] = gi;
      const float dx = ca[j][0] - ca[i][0], dy = ca[j][1] - ca[i][1], dz = ca[j][2] - ca[i][2];
      const float dca = 0.1f * norm_eps(dx, dy, dz);
      const float dcb = 0.1f * norm_eps(cb[j][0] - cb[i][0], cb[j][1] - cb[i][1], cb[j][2] - cb[i][2]);
      const double cx = (double)com[j][0] - (double)com[i][0], cy = (double)com[j][1] - (double)com[i][1], cz = (double)com[j][2] - (double)com[i][2];
      float* es = a.edge_s + (size_t)e * 21;
      es[0] = dca < 4.5f ? 1.f : 0.f;
      es[1] = norm_eps(dx, dy, dz) * 0.1f;                     // pairwise_distance(eps = 1e-6) * 0.1
      es[2] = (float)(sqrt(cx * cx + cy * cy + cz * cz) * 0.1);
      es[3] = dca; es[4] = dcb;
      for (int q = 0; q < 16; ++q) { const float u = (dca - (20.0f / 15.0f) * (float)q) / 1.25f; es[5 + q] = expf(-(u * u)); }
      unit3(dx, dy, dz, a.edge_v + (size_t)e * 3);
    }
  }
  if (g == gridDim.x - 1 && threadIdx.x == 0) a.in_ptr[a.n_res] = a.edge_ptr[gridDim.x];
}

extern "C" int dbfr_mdn_pocket_features(int32_t n_graph, int32_t n_res, const int32_t* res_ptr, const int32_t* edge_ptr,
                                        const int32_t* aatype, const float* atom14_pos, int32_t topk, float* node_s, float* node_v,
                                        int32_t* edge_src, int32_t* edge_dst, int32_t* in_ptr, float* edge_s, float* edge_v,
                                        void* hip_stream) {
  if (n_graph <= 0 || n_res <= 0 || !res_ptr || !edge_ptr || !aatype || !atom14_pos || !node_s || !node_v || !edge_src || !edge_dst ||
      !in_ptr || !edge_s || !edge_v) { dbfr_set_error("null argument"); return DBFR_ERR_ARG; }
  if (topk < 1 || topk > PF_MAXK) { dbfr_set_error("topk must be in [1, 32]"); return DBFR_ERR_ARG; }
  PocketFeatArgs a{res_ptr, edge_ptr, aatype, atom14_pos, topk, n_res, node_s, node_v, edge_src, edge_dst, in_ptr, edge_s, edge_v};
  hipLaunchKernelGGL(k_pocket_features, dim3(n_graph), dim3(128), 0, (hipStream_t)hip_stream, a);
  MCHECK(hipGetLastError());
  return DBFR_OK;
}
