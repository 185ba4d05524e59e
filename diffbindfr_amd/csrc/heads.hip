// Score heads and the on-device reverse-SDE geometry updates for gfx950.
//
// Replaces:
//   build_center_conv_graph + tr/rot magnitude MLPs + sigma scaling     tpscore.py:529-543,554-558,684-710
//   pseudotorque bond attributes, tor/sc_tor final tanh-MLPs             tpscore.py:546-571,716,741
//   Euler-Maruyama perturbations                                         scFlex.py:154-183,197-205
//   update_batchlig_pos -> modify_conformer (+Kabsch)                    conformer_utils.py:305-355,420-473
//   chi update + build_pdb_from_template                                 scFlex.py:207-226; aaframe.py:778-994
#include "common.h"

#define MAX_NL 256

__device__ __forceinline__ void vec_sh9(float vx, float vy, float vz, float* sh9, float* dist) {
  float nrm = sqrtf(vx * vx + vy * vy + vz * vz);
  float inv = 1.0f / fmaxf(nrm, 1e-12f);
  float x = vx * inv, y = vy * inv, z = vz * inv;
  const float s3 = 1.7320508075688772f, s15 = 3.872983346207417f, s5 = 2.23606797749979f;
  sh9[0] = 1.0f; sh9[1] = s3 * x; sh9[2] = s3 * y; sh9[3] = s3 * z;
  sh9[4] = s15 * x * z; sh9[5] = s15 * x * y; sh9[6] = s5 * (y * y - 0.5f * (x * x + z * z));
  sh9[7] = s15 * y * z; sh9[8] = (s15 * 0.5f) * (z * z - x * x);
  *dist = nrm;
}

// centre graph: one edge per ligand atom (tgt = graph, gth = atom), vec = pos - centroid
__global__ void k_center_edges(dbfr_batch b, int* tgt, int* gth, float* dist, float* sh, int* row_start, int* row_cnt) {
  const int g = blockIdx.x;
  const int l0 = b.lig_ptr[g], nl = b.lig_ptr[g + 1] - l0;
  __shared__ float c[3];
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int i = 0; i < nl; ++i) s += b.lig_pos[3 * (l0 + i) + threadIdx.x];   // index_add_ order
    c[threadIdx.x] = s / (float)nl;
  }
  if (threadIdx.x == 0) { row_start[g] = l0; row_cnt[g] = nl; }
  __syncthreads();
  for (int i = threadIdx.x; i < nl; i += blockDim.x) {
    int a = l0 + i;
    float s9[9], d;
    vec_sh9(b.lig_pos[3 * a] - c[0], b.lig_pos[3 * a + 1] - c[1], b.lig_pos[3 * a + 2] - c[2], s9, &d);
    tgt[a] = g; gth[a] = a; dist[a] = d;
    for (int k = 0; k < 9; ++k) sh[(size_t)a * SH_LD + k] = s9[k];
  }
}

void launch_center_edges(const dbfr_batch& b, int* tgt, int* gth, float* dist, float* sh, int* row_start, int* row_cnt,
                         hipStream_t st) {
  hipLaunchKernelGGL(k_center_edges, dim3(b.G), dim3(64), 0, st, b, tgt, gth, dist, sh, row_start, row_cnt);
}

// tr / rot heads: global_pred[G][12] = [1o_a 1o_b 1e_a 1e_b]
struct TrRotArgs {
  const float* gp; const float* temb; const float* tr_sigma; const float* rot_norm;
  Mlp2 tr, rot; int G; int scale_by_sigma; float* tr_out; float* rot_out; int* err;
};

// One hidden unit of the head MLP (tpscore.py:529-546: Linear(1 + 32 -> ns) - ReLU - Linear(ns -> 1)), times its output weight.
__device__ __forceinline__ float head_hidden(const Mlp2& w, int h, float nrm, const float* temb) {
  float a = w.b0 ? w.b0[h] : 0.f;
  a += w.w0t[h] * nrm;                               // w0t [in][hid], in index 0 = norm
  for (int i = 0; i < EMB; ++i) a += w.w0t[(1 + i) * w.hid + h] * temb[i];
  return w.w1t[h] * fmaxf(a, 0.f);                   // w1t [hid][1]
}

// A wavefront per graph: lane h evaluates hidden unit h of both heads (the weight rows are read coalesced), lane 0 adds the ns terms in
// index order (fixed order: reproducible).  The one-thread-per-graph form of rounds 1-3 took 64 us per step: 1.5 % of a predict.py-sized step.
__global__ __launch_bounds__(64) void k_trrot(TrRotArgs a) {
  __shared__ float part[2][64];
  const int g = blockIdx.x, h = threadIdx.x;
  const float* v = a.gp + (size_t)g * 12;
  float tr[3] = {v[0] + v[6], v[1] + v[7], v[2] + v[8]};
  float rt[3] = {v[3] + v[9], v[4] + v[10], v[5] + v[11]};
  float ntr = sqrtf(tr[0] * tr[0] + tr[1] * tr[1] + tr[2] * tr[2]);
  float nrt = sqrtf(rt[0] * rt[0] + rt[1] * rt[1] + rt[2] * rt[2]);
  part[0][h] = h < a.tr.hid ? head_hidden(a.tr, h, ntr, a.temb + g * EMB) : 0.f;
  part[1][h] = h < a.rot.hid ? head_hidden(a.rot, h, nrt, a.temb + g * EMB) : 0.f;
  __syncthreads();
  if (h != 0) return;
  float mtr = a.tr.b1 ? a.tr.b1[0] : 0.f, mrt = a.rot.b1 ? a.rot.b1[0] : 0.f;
  for (int q = 0; q < a.tr.hid; ++q) mtr += part[0][q];
  for (int q = 0; q < a.rot.hid; ++q) mrt += part[1][q];
  for (int k = 0; k < 3; ++k) {
    float t = tr[k] / ntr * mtr, r = rt[k] / nrt * mrt;
    if (a.scale_by_sigma) { t = t / a.tr_sigma[g]; r = r * a.rot_norm[g]; }
    if (!isfinite(t) || !isfinite(r)) atomicOr(a.err, 2);
    a.tr_out[3 * g + k] = t;
    a.rot_out[3 * g + k] = r;
  }
}

void launch_trrot(const TrRotArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_trrot, dim3(a.G), dim3(64), 0, st, a);      // (hid = ns = 48 <= 64: one lane per hidden unit)
}

// bond_attr[k][:NS] = x[b0][:NS] + x[b1][:NS]
__global__ void k_bond_attr(const float* x, int ldx, const int* b0, const int* b1, const int* bsel, int stride, int n,
                            float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * NS) return;
  int k = i / NS, c = i % NS;
  int u, v;
  if (bsel) { int e = bsel[k]; u = b0[e]; v = b1[e]; }
  else { u = b0[(size_t)k * stride]; v = b0[(size_t)k * stride + 1]; }
  out[i] = x[(size_t)u * ldx + c] + x[(size_t)v * ldx + c];
}

void launch_bond_attr(const float* x, int ldx, const int* b0, const int* b1, const int* bsel, int stride, int n,
                      float* out, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_bond_attr, dim3((n * NS + 255) / 256), dim3(256), 0, st, x, ldx, b0, b1, bsel, stride, n, out);
}

// torsion heads: tanh-MLP(96 -> 48 -> 1, no bias) * sqrt(score_norm2)
// tanh-MLP head of a torsion (tpscore.py:548-571): 2ns -> ns -> 1, no biases.  One hidden unit per thread (coalesced
// weight rows), the torsion's feature row shared through LDS, hidden units summed in a fixed order (reproducible).
#define TOR_PER_BLOCK 4
__global__ __launch_bounds__(TOR_PER_BLOCK * NS) void k_tor_final(const float* feat, Mlp2 w, const float* norm2, int scale, int n,
                                                                float* out) {
  __shared__ float f[TOR_PER_BLOCK][2 * NS];
  __shared__ float part[TOR_PER_BLOCK][NS];
  const int sub = threadIdx.x / NS, h = threadIdx.x - sub * NS;
  const int k = blockIdx.x * TOR_PER_BLOCK + sub;
  const bool live = k < n;
  for (int i = h; i < w.in; i += NS) f[sub][i] = live ? feat[(size_t)k * w.in + i] : 0.f;
  __syncthreads();
  float a = 0.f;
  for (int i = 0; i < w.in; ++i) a += w.w0t[i * w.hid + h] * f[sub][i];
  part[sub][h] = w.w1t[h] * tanhf(a);
  __syncthreads();
  if (h == 0 && live) {
    float o = 0.f;
    for (int q = 0; q < NS; ++q) o += part[sub][q];
    if (scale) o *= sqrtf(norm2[k]);
    out[k] = o;
  }
}

void launch_tor_final(const float* feat, const Mlp2& w, const float* norm2, int scale, int n, float* out,
                      hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_tor_final, dim3((n + TOR_PER_BLOCK - 1) / TOR_PER_BLOCK), dim3(TOR_PER_BLOCK * NS), 0, st, feat, w, norm2,
                     scale, n, out);
}

// ------------------------------------------------------------------------------------------------ SDE step
__device__ void axis_angle_to_rot(const float* aa, float* R) {
  // geometry_utils/utils.py:1056-1092 (quaternion, small-angle Taylor) + :672-720 (normalise, to matrix)
  float ang = sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
  float half = ang * 0.5f;
  float s = (fabsf(ang) < 1e-6f) ? (0.5f - (ang * ang) / 48.f) : (sinf(half) / ang);
  float q[4] = {cosf(half), aa[0] * s, aa[1] * s, aa[2] * s};
  float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float w = q[0] / nq, x = q[1] / nq, y = q[2] / nq, z = q[3] / nq;
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * x * y - 2 * w * z;         R[2] = 2 * x * z + 2 * w * y;
  R[3] = 2 * x * y + 2 * w * z;         R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * y * z - 2 * w * x;
  R[6] = 2 * x * z - 2 * w * y;         R[7] = 2 * y * z + 2 * w * x;         R[8] = w * w - x * x - y * y + z * z;
}

// rotation of the Kabsch problem: R = V diag(1,1,sign) U^T for H = U S V^T (one-sided Jacobi, fp64)
__device__ double kabsch_rotation(const double* Hin, double* R) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) A[i] = Hin[i];
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 3; ++i) { al += A[3 * i + p] * A[3 * i + p]; be += A[3 * i + q] * A[3 * i + q]; ga += A[3 * i + p] * A[3 * i + q]; }
        off += ga * ga;
        if (fabs(ga) <= 1e-300 || ga * ga <= 1e-32 * al * be) continue;
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          double ap = A[3 * i + p], aq = A[3 * i + q];
          A[3 * i + p] = c * ap - s * aq; A[3 * i + q] = s * ap + c * aq;
          double vp = V[3 * i + p], vq = V[3 * i + q];
          V[3 * i + p] = c * vp - s * vq; V[3 * i + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-60) break;
  }
  double sv[3], U[9];
  int kmin = 0;
  for (int k = 0; k < 3; ++k) {
    sv[k] = sqrt(A[k] * A[k] + A[3 + k] * A[3 + k] + A[6 + k] * A[6 + k]);
    if (sv[k] < sv[kmin]) kmin = k;
  }
  double smax = fmax(sv[0], fmax(sv[1], sv[2]));
  for (int k = 0; k < 3; ++k)
    for (int i = 0; i < 3; ++i) U[3 * i + k] = sv[k] > 1e-12 * smax ? A[3 * i + k] / sv[k] : 0.0;
  if (!(sv[kmin] > 1e-12 * smax)) {  // rank deficient: complete the basis
    int a = (kmin + 1) % 3, b = (kmin + 2) % 3;
    U[0 + kmin] = U[3 + a] * U[6 + b] - U[6 + a] * U[3 + b];
    U[3 + kmin] = U[6 + a] * U[0 + b] - U[0 + a] * U[6 + b];
    U[6 + kmin] = U[0 + a] * U[3 + b] - U[3 + a] * U[0 + b];
  }
  auto build = [&](double sgn) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double r = 0;
        for (int k = 0; k < 3; ++k) r += V[3 * i + k] * (k == kmin ? sgn : 1.0) * U[3 * j + k];
        R[3 * i + j] = r;
      }
    return R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
  };
  double det = build(1.0);
  if (det < 0) det = build(-1.0);
  return det;
}

struct SdeLigArgs {
  dbfr_batch b;
  const float* tr_score; const float* rot_score; const float* tor_score;
  const float* z_tr; const float* z_rot; const float* z_tor;     // this step's slices
  float dt, tr_g2, tr_gsdt, rot_g2, rot_gsdt, tor_g2, tor_gsdt;
  float* traj;   // [NL,3] slice or null
  int* err;
};

__device__ __forceinline__ float perturb(float g2, float score, float dt, float gsdt, float z) {
  return __fadd_rn(__fmul_rn(__fmul_rn(g2, score), dt), __fmul_rn(gsdt, z));
}

__global__ __launch_bounds__(128) void k_sde_ligand(SdeLigArgs a) {
  __shared__ float fx[MAX_NL], fy[MAX_NL], fz[MAX_NL], rx[MAX_NL], ry[MAX_NL], rz[MAX_NL];
  __shared__ float sh_c[3], sh_R[9], sh_t[3], sh_v[3];
  __shared__ double sh_H[9], sh_ca[3], sh_cb[3];
  const int g = blockIdx.x, tid = threadIdx.x;
  if (a.err[0] & 1) return;   // an edge list overflowed: the state stays at the beginning of that step (dbfr_capacity_report)
  const int l0 = a.b.lig_ptr[g], nl = a.b.lig_ptr[g + 1] - l0;
  const int k0 = a.b.tor_ptr[g], nt = a.b.tor_ptr[g + 1] - k0;
  for (int i = tid; i < nl; i += blockDim.x) {
    fx[i] = a.b.lig_pos[3 * (l0 + i)]; fy[i] = a.b.lig_pos[3 * (l0 + i) + 1]; fz[i] = a.b.lig_pos[3 * (l0 + i) + 2];
  }
  __syncthreads();
  if (tid < 3) {
    const float* p = tid == 0 ? fx : tid == 1 ? fy : fz;
    float s = 0.f;
    for (int i = 0; i < nl; ++i) s += p[i];
    sh_c[tid] = s / (float)nl;
    sh_t[tid] = perturb(a.tr_g2, a.tr_score[3 * g + tid], a.dt, a.tr_gsdt, a.z_tr[3 * g + tid]);
  }
  if (tid == 0) {
    float rp[3];
    for (int k = 0; k < 3; ++k) rp[k] = perturb(a.rot_g2, a.rot_score[3 * g + k], a.dt, a.rot_gsdt, a.z_rot[3 * g + k]);
    axis_angle_to_rot(rp, sh_R);
  }
  __syncthreads();
  for (int i = tid; i < nl; i += blockDim.x) {   // rigid = (x - c) R^T + tr + c
    float x = fx[i] - sh_c[0], y = fy[i] - sh_c[1], z = fz[i] - sh_c[2];
    float nx = (x * sh_R[0] + y * sh_R[1] + z * sh_R[2]) + sh_t[0] + sh_c[0];
    float ny = (x * sh_R[3] + y * sh_R[4] + z * sh_R[5]) + sh_t[1] + sh_c[1];
    float nz = (x * sh_R[6] + y * sh_R[7] + z * sh_R[8]) + sh_t[2] + sh_c[2];
    fx[i] = rx[i] = nx; fy[i] = ry[i] = ny; fz[i] = rz[i] = nz;
  }
  __syncthreads();
  if (nt > 0) {
    for (int k = 0; k < nt; ++k) {   // sequential, order dependent (conformer_utils.py:313-326)
      const float upd = perturb(a.tor_g2, a.tor_score[k0 + k], a.dt, a.tor_gsdt, a.z_tor[k0 + k]);
      if (upd == 0.f) continue;      // uniform over the block
      const int e = a.b.tor_bond[k0 + k];
      const int u = a.b.bond_src[e] - l0, v = a.b.bond_dst[e] - l0;
      if (tid == 0) {
        float ax[3] = {fx[u] - fx[v], fy[u] - fy[v], fz[u] - fz[v]};
        float nrm = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        float rv[3] = {ax[0] * upd / nrm, ax[1] * upd / nrm, ax[2] * upd / nrm};
        axis_angle_to_rot(rv, sh_R);
        sh_v[0] = fx[v]; sh_v[1] = fy[v]; sh_v[2] = fz[v];
      }
      __syncthreads();
      const uint8_t* mask = a.b.rot_mask + a.b.rot_mask_off[k0 + k];
      for (int i = tid; i < nl; i += blockDim.x)
        if (mask[i]) {
          float x = fx[i] - sh_v[0], y = fy[i] - sh_v[1], z = fz[i] - sh_v[2];
          fx[i] = (x * sh_R[0] + y * sh_R[1] + z * sh_R[2]) + sh_v[0];
          fy[i] = (x * sh_R[3] + y * sh_R[4] + z * sh_R[5]) + sh_v[1];
          fz[i] = (x * sh_R[6] + y * sh_R[7] + z * sh_R[8]) + sh_v[2];
        }
      __syncthreads();
    }
    // Kabsch: align flexible onto rigid (superimposition.py:375-410)
    if (tid == 0) {
      double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
      for (int i = 0; i < nl; ++i) { ca[0] += fx[i]; ca[1] += fy[i]; ca[2] += fz[i]; cb[0] += rx[i]; cb[1] += ry[i]; cb[2] += rz[i]; }
      for (int k = 0; k < 3; ++k) { ca[k] /= nl; cb[k] /= nl; sh_ca[k] = ca[k]; sh_cb[k] = cb[k]; }
      double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < nl; ++i) {
        double am[3] = {fx[i] - ca[0], fy[i] - ca[1], fz[i] - ca[2]};
        double bm[3] = {rx[i] - cb[0], ry[i] - cb[1], rz[i] - cb[2]};
        for (int p = 0; p < 3; ++p)
          for (int q = 0; q < 3; ++q) H[3 * p + q] += am[p] * bm[q];
      }
      double R[9];
      double det = kabsch_rotation(H, R);
      if (fabs(det - 1.0) >= 3e-3) atomicOr(a.err, 2);
      for (int k = 0; k < 9; ++k) sh_H[k] = R[k];
    }
    __syncthreads();
    for (int i = tid; i < nl; i += blockDim.x) {   // x R^T + t,  t = -R ca + cb
      double x = fx[i] - sh_ca[0], y = fy[i] - sh_ca[1], z = fz[i] - sh_ca[2];
      fx[i] = (float)(sh_H[0] * x + sh_H[1] * y + sh_H[2] * z + sh_cb[0]);
      fy[i] = (float)(sh_H[3] * x + sh_H[4] * y + sh_H[5] * z + sh_cb[1]);
      fz[i] = (float)(sh_H[6] * x + sh_H[7] * y + sh_H[8] * z + sh_cb[2]);
    }
    __syncthreads();
  }
  for (int i = tid; i < nl; i += blockDim.x) {
    float* o = a.b.lig_pos + 3 * (l0 + i);
    o[0] = fx[i]; o[1] = fy[i]; o[2] = fz[i];
    if (a.traj) { float* t = a.traj + 3 * (l0 + i); t[0] = fx[i]; t[1] = fy[i]; t[2] = fz[i]; }
  }
}

void launch_sde_ligand(const SdeLigArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_sde_ligand, dim3(a.b.G), dim3(128), 0, st, a);
}

// chi[mask] += perturb   (scFlex.py:208-210)
__global__ void k_sc_update(dbfr_batch b, const float* score, const float* z, float dt, float g2, float gsdt, const int* err) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= b.NSC || (err[0] & 1)) return;   // frozen after an edge-capacity overflow (see k_sde_ligand)
  int rc = b.sc_res_chi[k];
  int res = rc >> 2, chi = rc & 3;
  float* t = b.torsion_angle + (size_t)res * 5 + 1 + chi;
  *t = *t + perturb(g2, score[k], dt, gsdt, z[k]);
}

// side-chain rebuild: 8 rigid frames per residue -> atom14 -> compacted rec_pos
__global__ void k_atom14(dbfr_batch b, const int* a14_group /*[21][14]*/, float* atom14_out, float* traj14) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= b.NR) return;
  const float* ang = b.torsion_angle + (size_t)r * 5;
  // frames: 0 bb (identity), 3 psi, 4..7 chi1..4; (sin,cos) normalised with eps 1e-6 (msc.py:295-310)
  float Rf[8][9], tf[8][3];
  const float* DF = b.default_frame + (size_t)r * 128;
  for (int k = 0; k < 8; ++k) {
    float s, c;
    if (k == 0) { s = 0.f; c = 1.f; }
    else if (k < 3) { s = 0.f; c = 0.f; }
    else { s = sinf(ang[k - 3]); c = cosf(ang[k - 3]); }
    float nrm = fmaxf(sqrtf(s * s + c * c), 1e-6f);
    s /= nrm; c /= nrm;
    const float* D = DF + k * 16;   // 4x4 row-major
    // R = Rd * Rx(s,c),  Rx = [[1,0,0],[0,c,-s],[0,s,c]]
    for (int i = 0; i < 3; ++i) {
      float d0 = D[4 * i], d1 = D[4 * i + 1], d2 = D[4 * i + 2];
      Rf[k][3 * i] = d0; Rf[k][3 * i + 1] = d1 * c + d2 * s; Rf[k][3 * i + 2] = -d1 * s + d2 * c;
      tf[k][i] = D[4 * i + 3];
    }
  }
  for (int k = 5; k < 8; ++k) {   // chain chi2..4 onto the previous chi frame
    float Rn[9], tn[3];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j)
        Rn[3 * i + j] = Rf[k - 1][3 * i] * Rf[k][j] + Rf[k - 1][3 * i + 1] * Rf[k][3 + j] + Rf[k - 1][3 * i + 2] * Rf[k][6 + j];
      tn[i] = tf[k - 1][i] + (Rf[k - 1][3 * i] * tf[k][0] + Rf[k - 1][3 * i + 1] * tf[k][1] + Rf[k - 1][3 * i + 2] * tf[k][2]);
    }
    for (int i = 0; i < 9; ++i) Rf[k][i] = Rn[i];
    for (int i = 0; i < 3; ++i) tf[k][i] = tn[i];
  }
  const float* Rb = b.backbone_rots + (size_t)r * 9;
  const float* tb = b.backbone_transl + (size_t)r * 3;
  const int aa = b.sequence[r];
  for (int a = 0; a < 14; ++a) {
    const int slot = b.atom14_slot[(size_t)r * 14 + a];
    float out[3] = {0.f, 0.f, 0.f};
    if (slot >= 0) {
      const int k = a14_group[aa * 14 + a];
      const float* p = b.rigid_group_positions + ((size_t)r * 14 + a) * 3;
      // global frame = backbone o frame_k
      float Rg[9], tg[3];
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
          Rg[3 * i + j] = Rb[3 * i] * Rf[k][j] + Rb[3 * i + 1] * Rf[k][3 + j] + Rb[3 * i + 2] * Rf[k][6 + j];
        tg[i] = tb[i] + (Rb[3 * i] * tf[k][0] + Rb[3 * i + 1] * tf[k][1] + Rb[3 * i + 2] * tf[k][2]);
      }
      for (int i = 0; i < 3; ++i) out[i] = (Rg[3 * i] * p[0] + Rg[3 * i + 1] * p[1] + Rg[3 * i + 2] * p[2]) + tg[i];
      float* rp = b.rec_pos + (size_t)slot * 3;
      rp[0] = out[0]; rp[1] = out[1]; rp[2] = out[2];
    }
    if (atom14_out) { float* o = atom14_out + ((size_t)r * 14 + a) * 3; o[0] = out[0]; o[1] = out[1]; o[2] = out[2]; }
    if (traj14) { float* o = traj14 + ((size_t)r * 14 + a) * 3; o[0] = out[0]; o[1] = out[1]; o[2] = out[2]; }
  }
}

void launch_sidechain(const dbfr_batch& b, const float* score, const float* z, float dt, float g2, float gsdt,
                      const int* a14_group, float* atom14_out, float* traj14, const int* err, hipStream_t st) {
  if (b.NSC > 0) hipLaunchKernelGGL(k_sc_update, dim3((b.NSC + 255) / 256), dim3(256), 0, st, b, score, z, dt, g2, gsdt, err);
  if (b.NR > 0) hipLaunchKernelGGL(k_atom14, dim3((b.NR + 63) / 64), dim3(64), 0, st, b, a14_group, atom14_out, traj14);
}

// ------------------------------------------------------------------------------------------------ pose initialisation
// LigInit (struct_init.py:24-53): uniform torsion kicks applied in order (no Kabsch), then rotate about the centroid and
// translate -- the centroid is not added back.  One workgroup per ligand, the conformer lives in LDS.
__global__ __launch_bounds__(128) void k_init_ligand(dbfr_batch b, dbfr_init_tape z) {
  __shared__ float fx[MAX_NL], fy[MAX_NL], fz[MAX_NL];
  __shared__ float sh_R[9], sh_v[3], sh_c[3];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int l0 = b.lig_ptr[g], nl = b.lig_ptr[g + 1] - l0;
  const int k0 = b.tor_ptr[g], nt = b.tor_ptr[g + 1] - k0;
  for (int i = tid; i < nl; i += blockDim.x) {
    fx[i] = b.lig_pos[3 * (l0 + i)]; fy[i] = b.lig_pos[3 * (l0 + i) + 1]; fz[i] = b.lig_pos[3 * (l0 + i) + 2];
  }
  __syncthreads();
  for (int k = 0; k < nt; ++k) {
    const float upd = z.tor_u[k0 + k];
    if (upd == 0.f) continue;
    const int e = b.tor_bond[k0 + k];
    const int u = b.bond_src[e] - l0, v = b.bond_dst[e] - l0;
    if (tid == 0) {
      float ax[3] = {fx[u] - fx[v], fy[u] - fy[v], fz[u] - fz[v]};
      float nrm = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      float rv[3] = {ax[0] * upd / nrm, ax[1] * upd / nrm, ax[2] * upd / nrm};
      axis_angle_to_rot(rv, sh_R);
      sh_v[0] = fx[v]; sh_v[1] = fy[v]; sh_v[2] = fz[v];
    }
    __syncthreads();
    const uint8_t* mask = b.rot_mask + b.rot_mask_off[k0 + k];
    for (int i = tid; i < nl; i += blockDim.x)
      if (mask[i]) {
        float x = fx[i] - sh_v[0], y = fy[i] - sh_v[1], zc = fz[i] - sh_v[2];
        fx[i] = (x * sh_R[0] + y * sh_R[1] + zc * sh_R[2]) + sh_v[0];
        fy[i] = (x * sh_R[3] + y * sh_R[4] + zc * sh_R[5]) + sh_v[1];
        fz[i] = (x * sh_R[6] + y * sh_R[7] + zc * sh_R[8]) + sh_v[2];
      }
    __syncthreads();
  }
  if (tid < 3) {
    const float* p = tid == 0 ? fx : tid == 1 ? fy : fz;
    float s = 0.f;
    for (int i = 0; i < nl; ++i) s += p[i];
    sh_c[tid] = s / (float)nl;
  }
  if (tid < 9) sh_R[tid] = z.rot[9 * (size_t)g + tid];
  __syncthreads();
  for (int i = tid; i < nl; i += blockDim.x) {   // (x - c) R^T + tr
    float x = fx[i] - sh_c[0], y = fy[i] - sh_c[1], zc = fz[i] - sh_c[2];
    float* o = b.lig_pos + 3 * (size_t)(l0 + i);
    o[0] = (x * sh_R[0] + y * sh_R[1] + zc * sh_R[2]) + z.tr[3 * g];
    o[1] = (x * sh_R[3] + y * sh_R[4] + zc * sh_R[5]) + z.tr[3 * g + 1];
    o[2] = (x * sh_R[6] + y * sh_R[7] + zc * sh_R[8]) + z.tr[3 * g + 2];
  }
}

// SCProtInit (struct_init.py:117-124): chi <- draw * mask, psi untouched
__global__ void k_init_chi_zero(dbfr_batch b) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= b.NR) return;
  float* t = b.torsion_angle + (size_t)r * 5;
  t[1] = t[2] = t[3] = t[4] = 0.f;
}
__global__ void k_init_chi_set(dbfr_batch b, const float* sc_u) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= b.NSC) return;
  int rc = b.sc_res_chi[k];
  b.torsion_angle[(size_t)(rc >> 2) * 5 + 1 + (rc & 3)] = sc_u[rc];
}

void launch_init_poses(const dbfr_batch& b, const dbfr_init_tape& z, const int* a14_group, float* atom14_out, hipStream_t st) {
  hipLaunchKernelGGL(k_init_ligand, dim3(b.G), dim3(128), 0, st, b, z);
  hipLaunchKernelGGL(k_init_chi_zero, dim3((b.NR + 255) / 256), dim3(256), 0, st, b);
  if (b.NSC > 0) hipLaunchKernelGGL(k_init_chi_set, dim3((b.NSC + 255) / 256), dim3(256), 0, st, b, z.sc_u);
  hipLaunchKernelGGL(k_atom14, dim3((b.NR + 63) / 64), dim3(64), 0, st, b, a14_group, atom14_out, (float*)nullptr);
}

// ------------------------------------------------------------------------------------------------ pocket templates
// extract_chi_and_template (prot_math.py:116-241): the inverse of k_atom14.  One thread per residue: backbone frame from
// N, CA, C; psi and chi1..4 from the atom14 coordinates; per-residue default frames (the pose of every rigid group in
// its parent group) and the atoms' coordinates inside their groups.
#define RT_TABLE __device__ const
#include "residue_tables.inc"
#undef RT_TABLE

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 scale3(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }

struct Frame3 { V3 ex, ey, ez, t; };
// make_rigid_transformation_4x4 (geometry_utils/utils.py:93-122)
__device__ __forceinline__ Frame3 rigid_frame(V3 ex, V3 ey, V3 t) {
  Frame3 f;
  f.ex = scale3(ex, 1.f / (sqrtf(dot3(ex, ex)) + 1e-6f));
  V3 e = ey - scale3(f.ex, dot3(ey, f.ex));
  f.ey = scale3(e, 1.f / sqrtf(dot3(e, e)));
  f.ez = cross3(f.ex, f.ey);
  f.t = t;
  return f;
}
__device__ __forceinline__ V3 to_local(const Frame3& f, V3 p) {   // apply_inv_euclidean: R^T (p - t)
  V3 d = p - f.t;
  return v3(dot3(f.ex, d), dot3(f.ey, d), dot3(f.ez, d));
}
__device__ __forceinline__ void store_frame(float* m, const Frame3& f) {   // 4x4 row-major, columns (ex ey ez t)
  m[0] = f.ex.x; m[1] = f.ey.x; m[2] = f.ez.x; m[3] = f.t.x;
  m[4] = f.ex.y; m[5] = f.ey.y; m[6] = f.ez.y; m[7] = f.t.y;
  m[8] = f.ex.z; m[9] = f.ey.z; m[10] = f.ez.z; m[11] = f.t.z;
  m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
}

__global__ void k_extract_templates(int n_res, const int* __restrict__ aatype, const float* __restrict__ pos14,
                                    float* __restrict__ transl, float* __restrict__ rots, float* __restrict__ frames,
                                    float* __restrict__ rigid, float* __restrict__ angle) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_res) return;
  const int aa = aatype[r];
  V3 P[14];
  for (int a = 0; a < 14; ++a) { const float* p = pos14 + ((size_t)r * 14 + a) * 3; P[a] = v3(p[0], p[1], p[2]); }
  // residue_frame(origin = CA, x_axis = C, xy_plane = N)   (geometry_utils/utils.py:125-148)
  Frame3 bbf;
  {
    V3 e0 = P[2] - P[1], e1 = P[0] - P[1];
    e0 = scale3(e0, 1.f / sqrtf(dot3(e0, e0) + 1e-20f));
    e1 = e1 - scale3(e0, dot3(e0, e1));
    e1 = scale3(e1, 1.f / sqrtf(dot3(e1, e1) + 1e-20f));
    bbf.ex = e0; bbf.ey = e1; bbf.ez = cross3(e0, e1); bbf.t = P[1];
  }
  transl[3 * r] = P[1].x; transl[3 * r + 1] = P[1].y; transl[3 * r + 2] = P[1].z;
  {
    float* R = rots + (size_t)r * 9;       // columns e0 e1 e2
    R[0] = bbf.ex.x; R[1] = bbf.ey.x; R[2] = bbf.ez.x;
    R[3] = bbf.ex.y; R[4] = bbf.ey.y; R[5] = bbf.ez.y;
    R[6] = bbf.ex.z; R[7] = bbf.ey.z; R[8] = bbf.ez.z;
  }
  V3 cur[14], tm[14];
  for (int a = 0; a < 14; ++a) { cur[a] = to_local(bbf, P[a]); tm[a] = v3(0.f, 0.f, 0.f); }
  tm[0] = v3(cur[0].x, cur[0].y, 0.f);      // N: x, y
  tm[2] = v3(cur[2].x, 0.f, 0.f);           // C: x
  tm[4] = cur[4];                           // CB
  float* F = frames + (size_t)r * 128;
  for (int i = 0; i < 128; ++i) F[i] = 0.f;
  F[0] = F[5] = F[10] = F[15] = 1.f;                    // backbone group
  F[16] = F[21] = F[26] = F[31] = 1.f;                  // pre-omega group (empty)
  store_frame(F + 32, rigid_frame(tm[0] - tm[1], v3(1.f, 0.f, 0.f), tm[0]));          // phi group (empty)
  const Frame3 psi_f = rigid_frame(tm[2] - tm[1], tm[1] - tm[0], tm[2]);
  store_frame(F + 48, psi_f);
  float* A = angle + (size_t)r * 5;
  {
    const V3 o = to_local(psi_f, cur[3]);               // O in the psi frame -> (x, |yz|, 0), psi = atan2(z, y)
    tm[3] = v3(o.x, sqrtf(o.y * o.y + o.z * o.z), 0.f);
    A[0] = atan2f(o.z, o.y);
  }
  for (int k = 0; k < 4; ++k) {
    A[k + 1] = 0.f;
    if (!kChiMask[aa][k]) continue;
    const int i0 = kChiAtoms14[aa][k][0], i1 = kChiAtoms14[aa][k][1], i2 = kChiAtoms14[aa][k][2], i3 = kChiAtoms14[aa][k][3];
    const Frame3 f = k == 0 ? rigid_frame(cur[i2] - cur[i1], cur[i0] - cur[i1], cur[i2])
                            : rigid_frame(cur[i2], v3(-1.f, 0.f, 0.f), cur[i2]);
    store_frame(F + 16 * (4 + k), f);
    const V3 last = to_local(f, cur[i3]);
    const float chi = atan2f(last.z, last.y);
    A[k + 1] = chi;
    const float sn = sinf(-chi), cs = cosf(-chi);       // rot_vec_around_x_axis(x, -chi)
    for (int a = 0; a < 14; ++a) {
      const V3 l = to_local(f, cur[a]);
      cur[a] = v3(l.x, cs * l.y - sn * l.z, sn * l.y + cs * l.z);
      if (kAtom14ToGroup[aa * 14 + a] == k + 4) { tm[a].x += cur[a].x; tm[a].y += cur[a].y; tm[a].z += cur[a].z; }
    }
  }
  // * restype_atom14_mask (ideal mask of the residue type): the slots the type does not use stay zero
  for (int a = 0; a < 14; ++a) {
    const bool used = kAtom14Mask[aa][a] != 0;
    float* o = rigid + ((size_t)r * 14 + a) * 3;
    o[0] = used ? tm[a].x : 0.f; o[1] = used ? tm[a].y : 0.f; o[2] = used ? tm[a].z : 0.f;
  }
}

void launch_extract_templates(int n_res, const int* aatype, const float* pos14, float* transl, float* rots, float* frames,
                              float* rigid, float* angle, hipStream_t st) {
  if (n_res > 0)
    hipLaunchKernelGGL(k_extract_templates, dim3((n_res + 63) / 64), dim3(64), 0, st, n_res, aatype, pos14, transl, rots,
                       frames, rigid, angle);
}

// ------------------------------------------------------------------------------------------------ pocket residue selection
// select_bs (druglib/utils/bio_utils/select_pocket.py:12-99) for any number of proteins in one pass: thread per residue finds
// its protein by bisection in res_ptr, then the minimum squared distance between its present atoms and the protein's
// reference points (float32, summed x, y, z in that order like the reference's torch.sum; absent atoms count as 1e20).
__global__ void k_pocket_min_dist(int n_prot, int n_res_total, const int* __restrict__ res_ptr, int m, const float* __restrict__ pos,
                                  const float* __restrict__ mask, const int* __restrict__ ref_ptr, const float* __restrict__ ref,
                                  float* __restrict__ d2out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_res_total) return;
  int lo = 0, hi = n_prot;                       // res_ptr[lo] <= r < res_ptr[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (res_ptr[mid] <= r) lo = mid; else hi = mid;
  }
  const int l0 = ref_ptr[lo], l1 = ref_ptr[lo + 1];
  float best = 1e20f;
  for (int a = 0; a < m; ++a) {
    if (mask[(size_t)r * m + a] == 0.f) continue;
    const float x = pos[((size_t)r * m + a) * 3], y = pos[((size_t)r * m + a) * 3 + 1], z = pos[((size_t)r * m + a) * 3 + 2];
    for (int l = l0; l < l1; ++l) {
      const float dx = x - ref[3 * l], dy = y - ref[3 * l + 1], dz = z - ref[3 * l + 2];
      best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
  }
  d2out[r] = best;
}

// One workgroup per protein: the nearest residue is always kept (first index on ties, like torch.argmin); with a neighbour
// cap only the max_neighbors nearest selected residues stay (ties by index).
__global__ __launch_bounds__(256) void k_pocket_finalize(const int* __restrict__ res_ptr, const float* __restrict__ d2, float cut2,
                                                         int max_neighbors, unsigned char* __restrict__ out) {
  __shared__ float sd[256];
  __shared__ int si[256];
  const int r0 = res_ptr[blockIdx.x], n = res_ptr[blockIdx.x + 1] - r0;
  const int tid = threadIdx.x;
  float bd = INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < n; i += 256) {
    float v = d2[r0 + i];
    if (v < bd) { bd = v; bi = i; }              // ascending i per thread: keeps the first minimum
  }
  sd[tid] = bd; si[tid] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      if (sd[tid + s] < sd[tid] || (sd[tid + s] == sd[tid] && si[tid + s] < si[tid])) { sd[tid] = sd[tid + s]; si[tid] = si[tid + s]; }
    }
    __syncthreads();
  }
  const int amin = si[0];
  for (int i = tid; i < n; i += 256) {
    const float v = d2[r0 + i];
    bool in = v <= cut2 || i == amin;
    if (in && max_neighbors > 0) {
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const float w = d2[r0 + j];
        const bool jin = w <= cut2 || j == amin;
        rank += (jin && (w < v || (w == v && j < i))) ? 1 : 0;
      }
      in = rank < max_neighbors;
    }
    out[r0 + i] = in ? 1 : 0;
  }
}

void launch_select_pocket(int n_prot, int n_res_total, const int* res_ptr, int m, const float* pos, const float* mask, const int* ref_ptr,
                          const float* ref, float cut2, int max_neighbors, float* d2, unsigned char* out, hipStream_t st) {
  if (n_prot <= 0 || n_res_total <= 0) return;
  hipLaunchKernelGGL(k_pocket_min_dist, dim3((n_res_total + 127) / 128), dim3(128), 0, st, n_prot, n_res_total, res_ptr, m, pos, mask,
                     ref_ptr, ref, d2);
  hipLaunchKernelGGL(k_pocket_finalize, dim3(n_prot), dim3(256), 0, st, res_ptr, d2, cut2, max_neighbors, out);
}

// ------------------------------------------------------------------------------------------------ small utilities
__global__ void k_fill(float* p, float v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill(float* p, float v, int n, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, st, p, v, n);
}
__global__ void k_set_int(int* p, int v) { *p = v; }
void launch_set_int(int* p, int v, hipStream_t st) { hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, p, v); }
__global__ void k_acc_flops(const int* n_edges, double flops_per_edge, double bytes_per_edge, double fused_bytes_per_edge, double* counter) {
  // atomics: in profile mode 2 the four convs of a layer run on four streams
  atomicAdd(&counter[0], flops_per_edge * (double)*n_edges);
  atomicAdd(&counter[1], bytes_per_edge * (double)*n_edges);
  atomicAdd(&counter[2], fused_bytes_per_edge * (double)*n_edges);
}
__global__ void k_acc_executed(const int* n_edges, double flops_per_edge, double* counter) { atomicAdd(counter, flops_per_edge * (double)*n_edges); }
void launch_acc_executed(const int* n_edges, double flops_per_edge, double* counter, hipStream_t st) {
  hipLaunchKernelGGL(k_acc_executed, dim3(1), dim3(1), 0, st, n_edges, flops_per_edge, counter);
}
// (profiling) all the counter updates of one conv launch in ONE kernel: a launch per update was 66 one-thread launches per denoise step, ~1 % of the
// step the profile is taken of
struct AccBatch { const int* n[32]; double coef[32]; double* dst[32]; int count; };
__global__ void k_acc_batch(AccBatch b) {
  const int i = threadIdx.x;
  if (i < b.count) atomicAdd(b.dst[i], b.coef[i] * (double)*b.n[i]);
}
void launch_acc_batch(const AccBatch& b, hipStream_t st) {
  if (b.count > 0) hipLaunchKernelGGL(k_acc_batch, dim3(1), dim3(32), 0, st, b);
}
void launch_acc_flops(const int* n_edges, double flops_per_edge, double bytes_per_edge, double fused_bytes_per_edge, double* counter, hipStream_t st) {
  hipLaunchKernelGGL(k_acc_flops, dim3(1), dim3(1), 0, st, n_edges, flops_per_edge, bytes_per_edge, fused_bytes_per_edge, counter);
}
