// Output side of the sampler: per-pose metrics on the device + PDB text on the host.
// Replaces what complex_modeling (DiffBindFR/evaluation/export.py:106-312) does per pose in Python.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dbfr.h"
#include "common.h"

#define RT_TABLE __device__ const
#include "residue_tables.inc"
#undef RT_TABLE

// ------------------------------------------------------------------------------------------------ device: metrics
// A workgroup of 256 threads owns `ppb` consecutive (pose, frame) pairs -- as many as fit PM_ROWS = 64 pocket rows -- or
// one pair walked in 64-row tiles when the pocket is larger.  Pocket rows of the workgroup's pairs are contiguous in the
// trajectory: staged through LDS with coalesced loads (168 B per row), then FOUR threads per residue row: thread k of a
// row owns chi_k and a quarter of the side-chain slots, the quad combines with DPP.  Static per-row data (chi atom
// slots, masks, the other naming's slots, target chi angles) is built once per tile into LDS and shared by the pairs.
// Ligand poses sit in LDS for the automorphism loop: one lane per (pair, automorphism) for ligands up to 128 atoms,
// one wave per automorphism above.  All reductions run in a fixed order (bitwise reproducible).  The kernel reads every
// trajectory byte exactly once: HBM-bound by construction, algorithmic bytes = 12 * (n_lig + 14 n_res) per (pose, frame);
// targets, masks and tables stay cache resident.  The ligand half and the pocket half run as two launches (different
// pairs per workgroup); the pocket launch holds 15 KB of LDS per workgroup => 8 workgroups (32 waves) per CU.
#define PM_THREADS 256
#define PM_ROWS 64               // pocket rows per tile (4 threads each)
#define PM_MAX_PPB 25
#define PM_LIG_LDS 3072          // floats of ligand coordinates in LDS
#define PM_SMALL_LIG 128         // up to here: one lane per (pair, automorphism) instead of one wave
#define PM_MAX_LIG (PM_LIG_LDS / 3)

struct PmArgs {
  dbfr_pose_metrics_in in;
  dbfr_pose_metrics_out out;
  int ppb;                       // (pose, frame) pairs per workgroup
  int small;                     // ligand path: 1 = lane per (pair, automorphism)
  long long n_pf;
};

__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum over the 4 lanes of a quad (DPP quad_perm [1,0,3,2] then [2,3,0,1]); every lane gets the same value
__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
  return v;
}

struct P3 { float x, y, z; };
__device__ __forceinline__ P3 sub3(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dotp(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// (sin, cos) of the dihedral a0-a1-a2-a3 the way AlphaFold's atom37_to_torsion_angles builds it: frame with origin a2,
// x along a2-a1, a0 in the xy plane (eps 1e-8 under the roots), a3 expressed in it, (z, y) normalised.
__device__ __forceinline__ void chi_sin_cos(P3 a0, P3 a1, P3 a2, P3 a3, float& s, float& c) {
  P3 e0 = sub3(a2, a1), e1 = sub3(a0, a2);
  float inv = 1.f / sqrtf(dotp(e0, e0) + 1e-8f);
  e0 = {e0.x * inv, e0.y * inv, e0.z * inv};
  float d = dotp(e0, e1);
  e1 = {e1.x - e0.x * d, e1.y - e0.y * d, e1.z - e0.z * d};
  inv = 1.f / sqrtf(dotp(e1, e1) + 1e-8f);
  e1 = {e1.x * inv, e1.y * inv, e1.z * inv};
  P3 e2 = {e0.y * e1.z - e0.z * e1.y, e0.z * e1.x - e0.x * e1.z, e0.x * e1.y - e0.y * e1.x};
  P3 rel = sub3(a3, a2);
  s = dotp(e2, rel);
  c = dotp(e1, rel);
  inv = 1.f / sqrtf(s * s + c * c + 1e-8f);
  s *= inv;
  c *= inv;
}

// metrics/angbin.py:11-29 with its argument order at :98-99 (fmod keeps the dividend's sign)
__device__ __forceinline__ float ang_diff(float ang_pred, float ang_tgt) {
  const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
  // x = ang_tgt - ang_pred + pi lies in [-pi, 3 pi]: fmod(x, 2 pi) is x - 2 pi above 2 pi (exact, Sterbenz) and x otherwise
  float x = ang_tgt - ang_pred + pi;
  float d = (x >= two_pi ? x - two_pi : x) - pi;
  return fminf(fmaxf(fabsf(d), 0.f), pi);
}

// Static description of one pocket row, built once per tile and shared by the workgroup's pairs (LDS)
struct RowInfo {
  unsigned short chi[4];     // atom14 slots of the chi_k dihedral, 4 bits each
  unsigned char flag[4];     // bit 0: chi_k exists (type has it and its 4 atoms are present), bit 1: pi-periodic
  unsigned mask14;           // atoms present
  unsigned swap_lo;          // atom14 slot of the other naming for slots 5..12, 4 bits each
  unsigned swap_hi;          // ... and for slot 13
  float ang[4];              // chi_k of the target structure
};

__global__ __launch_bounds__(PM_THREADS) void k_pose_metrics(PmArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[PM_ROWS * 42];         // pose rows; reused for per-thread partials
  extern __shared__ float lig[];                                            // PM_LIG_LDS floats in the ligand launch, none in the pocket launch
  __shared__ RowInfo info[PM_ROWS];
  __shared__ float sums[PM_MAX_PPB][10];
  __shared__ float best_w[PM_MAX_PPB][PM_THREADS / 64];
  __shared__ float cen[PM_MAX_PPB + 1][3];
  __shared__ unsigned char hvy[PM_SMALL_LIG];
  const dbfr_pose_metrics_in& in = a.in;
  const long long pf0 = (long long)blockIdx.x * a.ppb;
  const int npf = (int)min((long long)a.ppb, a.n_pf - pf0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float cx = in.center[0], cy = in.center[1], cz = in.center[2];

  // ---- ligand: centroid distance and symmetry-corrected RMSD
  if (a.out.centroid || a.out.lig_rmsd) {
    const int nl3 = in.n_lig * 3;
    const float* src = in.lig_traj + (size_t)pf0 * nl3;
    for (int i = tid; i < npf * nl3; i += PM_THREADS) {
      int c = i % 3;
      lig[i] = src[i] + (c == 0 ? cx : (c == 1 ? cy : cz));
    }
    float* ligt = lig + npf * nl3;                       // small ligands: target + heavy-atom flags next to the poses
    if (a.small) {
      for (int i = tid; i < nl3; i += PM_THREADS) ligt[i] = in.lig_target[i];
      for (int i = tid; i < in.n_lig; i += PM_THREADS) hvy[i] = !in.heavy_mask || in.heavy_mask[i];
    }
    __syncthreads();
    if (a.small) {
      // one lane per (pair, coordinate) / per (pair, automorphism), serial over atoms -- no cross-lane traffic
      if (a.out.centroid && tid < (npf + 1) * 3) {
        const int q = tid / 3, c = tid - 3 * q;
        const float* lp = lig + q * nl3;                                     // row npf: the target's centroid
        float sacc = 0.f;
        for (int i = 0; i < in.n_lig; ++i) sacc += lp[3 * i + c];
        cen[q][c] = sacc / (float)in.n_lig;
      }
      if (a.out.lig_rmsd) {
        for (int it = tid; it < npf * in.n_perm; it += PM_THREADS) {
          const int q = it / in.n_perm, p = it - q * in.n_perm;
          const float* lp = lig + q * nl3;
          const int* perm = in.perms + (size_t)p * in.n_lig;
          float acc = 0.f, cnt = 0.f;
#pragma unroll 4
          for (int i = 0; i < in.n_lig; ++i) {
            const int j = perm[i];
            const float dx = lp[3 * j] - ligt[3 * i], dy = lp[3 * j + 1] - ligt[3 * i + 1], dz = lp[3 * j + 2] - ligt[3 * i + 2];
            if (hvy[i] && hvy[j]) { acc += dx * dx + dy * dy + dz * dz; cnt += 1.f; }
          }
          tile[it] = sqrtf(acc / cnt);                   // npf * n_perm <= PM_ROWS * 42 values; min taken below
        }
      }
      __syncthreads();
      if (tid < npf) {
        if (a.out.centroid) {
          float dx = cen[tid][0] - cen[npf][0], dy = cen[tid][1] - cen[npf][1], dz = cen[tid][2] - cen[npf][2];
          a.out.centroid[pf0 + tid] = sqrtf(dx * dx + dy * dy + dz * dz);
        }
        if (a.out.lig_rmsd) {
          float b = INFINITY;
          for (int p = 0; p < in.n_perm; ++p) b = fminf(b, tile[tid * in.n_perm + p]);
          a.out.lig_rmsd[pf0 + tid] = b;
        }
      }
      __syncthreads();
    } else {
      // large ligands / very many automorphisms: waves over automorphisms, lanes over atoms
      for (int q = 0; q < npf; ++q) {
        const float* lp = lig + q * nl3;
        if (a.out.centroid && wave == (q & 3)) {
          float sx = 0, sy = 0, sz = 0, tx = 0, ty = 0, tz = 0;
          for (int i = lane; i < in.n_lig; i += 64) {
            sx += lp[3 * i]; sy += lp[3 * i + 1]; sz += lp[3 * i + 2];
            tx += in.lig_target[3 * i]; ty += in.lig_target[3 * i + 1]; tz += in.lig_target[3 * i + 2];
          }
          sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
          tx = wave_sum(tx); ty = wave_sum(ty); tz = wave_sum(tz);
          if (lane == 0) {
            float n = (float)in.n_lig;
            float dx = sx / n - tx / n, dy = sy / n - ty / n, dz = sz / n - tz / n;
            a.out.centroid[pf0 + q] = sqrtf(dx * dx + dy * dy + dz * dz);
          }
        }
        if (a.out.lig_rmsd) {
          float best = INFINITY;
          for (int p = wave; p < in.n_perm; p += PM_THREADS / 64) {
            const int* perm = in.perms + (size_t)p * in.n_lig;
            float acc = 0.f, cnt = 0.f;
            for (int i = lane; i < in.n_lig; i += 64) {
              int j = perm[i];
              if (!in.heavy_mask || (in.heavy_mask[i] && in.heavy_mask[j])) {
                float dx = lp[3 * j] - in.lig_target[3 * i], dy = lp[3 * j + 1] - in.lig_target[3 * i + 1],
                      dz = lp[3 * j + 2] - in.lig_target[3 * i + 2];
                acc += dx * dx + dy * dy + dz * dz;
                cnt += 1.f;
              }
            }
            acc = wave_sum(acc);
            cnt = wave_sum(cnt);
            best = fminf(best, sqrtf(acc / cnt));      // lane 0 holds the sums
          }
          if (lane == 0) best_w[q][wave] = best;
        }
      }
      if (a.out.lig_rmsd) {
        __syncthreads();
        if (tid < npf) {
          float b = best_w[tid][0];
          for (int w = 1; w < PM_THREADS / 64; ++w) b = fminf(b, best_w[tid][w]);
          a.out.lig_rmsd[pf0 + tid] = b;
        }
      }
    }
  }
  if (!(a.out.sc_rmsd || a.out.chi_rate || a.out.delta_chi)) return;

  // ---- pocket: side-chain RMSD and chi differences, four threads per residue row
  const int rl = tid >> 2, k = tid & 3;                  // row inside the tile, chi / slot quarter of this thread
  float v_rmsd = 0.f, v_cnt = 0.f, v_ok = 0.f, v_ex = 0.f;
  const int segr = a.ppb > 1 ? in.n_res : PM_ROWS;        // rows of one pair inside the workgroup's tile
  const float* psrc = in.prot_traj + (size_t)pf0 * in.n_res * 42;
  for (int r0 = 0; r0 < (a.ppb > 1 ? 1 : in.n_res); r0 += PM_ROWS) {
    const int trows = a.ppb > 1 ? in.n_res : min(PM_ROWS, in.n_res - r0);      // distinct target rows in this pass
    const int rows = a.ppb > 1 ? npf * in.n_res : trows;                        // pose rows in this pass
    __syncthreads();
    {   // rows * 42 floats, contiguous; float2 keeps every row start aligned (42 is even)
      const float2* s2 = reinterpret_cast<const float2*>(psrc + (size_t)r0 * 42);
      float2* d2 = reinterpret_cast<float2*>(tile);
      for (int i = tid; i < rows * 21; i += PM_THREADS) d2[i] = s2[i];
    }
    if (rl < trows) {                                     // static row data: thread k fills the chi_k fields
      const int r = r0 + rl;
      const int aa = in.aatype[r];
      const float* tg = in.atom14_target + (size_t)r * 42;
      const float* mk = in.atom14_target_mask + (size_t)r * 14;
      unsigned mask14 = 0;
      for (int j = 0; j < 14; ++j) mask14 |= (mk[j] != 0.f ? 1u : 0u) << j;
      float ang = 0.f;
      unsigned short packed = 0;
      unsigned char flag = 0;
      if (kChiMask[aa][k]) {
        const int* ia = kChiAtoms14[aa][k];
        P3 t4[4];
        bool all = true;
        for (int u = 0; u < 4; ++u) {
          t4[u] = {tg[3 * ia[u]], tg[3 * ia[u] + 1], tg[3 * ia[u] + 2]};
          packed |= (unsigned short)(ia[u] << (4 * u));
          all = all && ((mask14 >> ia[u]) & 1u);
        }
        float ts, tc;
        chi_sin_cos(t4[0], t4[1], t4[2], t4[3], ts, tc);
        ang = atan2f(ts, tc);
        flag = (all ? 1 : 0) | (kChiPiPeriodic[aa][k] ? 2 : 0);
      }
      info[rl].chi[k] = packed;
      info[rl].flag[k] = flag;
      info[rl].ang[k] = ang;
      if (k == 0) {
        unsigned lo = 0;
        for (int j = 5; j < 13; ++j) lo |= (unsigned)kAtom14Swap[aa][j] << (4 * (j - 5));
        info[rl].mask14 = mask14;
        info[rl].swap_lo = lo;
        info[rl].swap_hi = (unsigned)kAtom14Swap[aa][13];
      }
    }
    __syncthreads();
    if (rl < rows) {
      const int q = a.ppb > 1 ? rl / in.n_res : 0;
      const int rt = a.ppb > 1 ? rl - q * in.n_res : rl;    // row inside the pair's tile = index into info
      const int r = r0 + rt;
      const unsigned mask14 = info[rt].mask14, swap_lo = info[rt].swap_lo, swap_hi = info[rt].swap_hi;
      const float* pr = tile + rl * 42;
      const float* tg = in.atom14_target + (size_t)r * 42;
      // side chain: slots 5..13 dealt to the quad (k: 5+k, 9+k, and 13 to k == 0), both namings
      float d2 = 0.f, d2a = 0.f;
      for (int t = 0; t < 3; ++t) {
        const int j = t < 2 ? 5 + 4 * t + k : 13;
        if (t == 2 && k != 0) break;
        const int js = j < 13 ? (int)((swap_lo >> (4 * (j - 5))) & 15u) : (int)swap_hi;
        const float m = (float)((mask14 >> j) & 1u), ma = (float)((mask14 >> js) & 1u);
        for (int c = 0; c < 3; ++c) {
          float cc = c == 0 ? cx : (c == 1 ? cy : cz);
          float p = (pr[3 * j + c] + cc) * m;
          float tt = (tg[3 * j + c] + cc) * m;
          float ta = (tg[3 * js + c] + cc) * ma;
          d2 += (tt - p) * (tt - p);
          d2a += (ta - p) * (ta - p);
        }
      }
      d2 = quad_sum(d2);
      d2a = quad_sum(d2a);
      const unsigned scbits = (mask14 >> 5) & 0x1ffu;
      if (k == 0 && scbits) {
        v_rmsd += sqrtf(fminf(d2, d2a) / ((float)__popc(scbits) + 1e-6f));
        v_cnt += 1.f;
      }
      // chi_k
      float dl = 0.f;
      const unsigned flag = info[rt].flag[k];
      if (flag & 1u) {
        const unsigned pk = info[rt].chi[k];
        P3 p4[4];
        for (int u = 0; u < 4; ++u) {
          const int s14 = (pk >> (4 * u)) & 15;
          p4[u] = {pr[3 * s14] + cx, pr[3 * s14 + 1] + cy, pr[3 * s14 + 2] + cz};
        }
        float ps, pc;
        chi_sin_cos(p4[0], p4[1], p4[2], p4[3], ps, pc);
        const float pi = 3.14159265358979323846f;
        const float ap = atan2f(ps, pc), at = info[rt].ang[k];
        dl = ang_diff(ap, at);
        if (flag & 2u) dl = fminf(dl, ang_diff(ap, at > 0.f ? at - pi : at + pi));     // atan2(-sin, -cos)
        v_ex += 1.f;
        if (dl < in.chi_bound) v_ok += 1.f;
      }
      if (a.out.delta_chi) a.out.delta_chi[((size_t)(pf0 + q) * in.n_res + r) * 4 + k] = dl;
    }
  }
  // ---- fixed-order reduction of the per-thread partials over each pair's rows (partials overlay the pose tile)
  __syncthreads();
  float* part = tile;                                  // [PM_THREADS][5]
  part[tid * 5 + 0] = v_rmsd; part[tid * 5 + 1] = v_cnt; part[tid * 5 + 2] = v_ok; part[tid * 5 + 3] = v_ex;
  __syncthreads();
  if (tid < npf * 10) {
    const int q = tid / 10, w = tid - q * 10;          // w: 0 rmsd, 1 rows with a side chain, 2..5 chi ok, 6..9 chi exists
    const int kk = w < 2 ? 0 : (w - 2) & 3, slot = w < 2 ? w : (w < 6 ? 2 : 3);
    float sacc = 0.f;
    for (int i = 0; i < segr; ++i) sacc += part[((q * segr + i) * 4 + kk) * 5 + slot];
    sums[q][w] = sacc;
  }
  __syncthreads();
  if (tid < npf) {
    if (a.out.sc_rmsd) a.out.sc_rmsd[pf0 + tid] = sums[tid][0] / sums[tid][1];
    if (a.out.chi_rate)
      for (int c = 0; c < 4; ++c) a.out.chi_rate[(size_t)(pf0 + tid) * 4 + c] = sums[tid][2 + c] / sums[tid][6 + c];
  }
}

extern "C" int dbfr_pose_metrics(const dbfr_pose_metrics_in* in, const dbfr_pose_metrics_out* out, void* hip_stream) {
  if (!in || !out) { dbfr_set_error("null argument"); return DBFR_ERR_ARG; }
  if (in->n_pose < 0 || in->n_frame < 0 || in->n_lig < 0 || in->n_res < 0) { dbfr_set_error("negative size"); return DBFR_ERR_ARG; }
  if ((long long)in->n_pose * in->n_frame == 0) return DBFR_OK;          // nothing to do (empty tensors may carry null pointers)
  const bool want_lig = out->centroid || out->lig_rmsd, want_prot = out->sc_rmsd || out->chi_rate || out->delta_chi;
  if (want_lig && (!in->lig_traj || !in->lig_target || in->n_lig < 1)) { dbfr_set_error("ligand metrics need lig_traj / lig_target"); return DBFR_ERR_ARG; }
  if (want_lig && in->n_lig > PM_MAX_LIG) { dbfr_set_error("ligand larger than 1024 atoms"); return DBFR_ERR_ARG; }
  if (out->lig_rmsd && (in->n_perm < 1 || !in->perms)) { dbfr_set_error("lig_rmsd needs at least the identity automorphism"); return DBFR_ERR_ARG; }
  if (want_prot && (!in->prot_traj || !in->atom14_target || !in->atom14_target_mask || !in->aatype || in->n_res < 1)) {
    dbfr_set_error("pocket metrics need prot_traj / atom14_target / atom14_target_mask / aatype");
    return DBFR_ERR_ARG;
  }
  if (want_prot && ((uintptr_t)in->prot_traj & 7)) { dbfr_set_error("prot_traj must be 8-byte aligned"); return DBFR_ERR_ARG; }
  const long long n_pf = (long long)in->n_pose * in->n_frame;
  if (n_pf == 0 || !(want_lig || want_prot)) return DBFR_OK;
  // Two launches of the same kernel: the ligand half and the pocket half want different numbers of pairs per workgroup
  // (a workgroup holds up to 25 ligand poses but only 64 pocket rows), and sharing one value starves the ligand lanes.
  for (int half = 0; half < 2; ++half) {
    if (half == 0 ? !want_lig : !want_prot) continue;
    PmArgs a;
    a.in = *in;
    a.out = *out;
    a.n_pf = n_pf;
    a.small = 0;
    int ppb = PM_MAX_PPB;
    if (half == 0) {
      a.out.sc_rmsd = a.out.chi_rate = a.out.delta_chi = nullptr;
      const int n_perm = out->lig_rmsd ? in->n_perm : 1;
      a.small = in->n_lig <= PM_SMALL_LIG && n_perm <= PM_ROWS * 42 && 2 * 3 * in->n_lig <= PM_LIG_LDS;
      if (a.small) {
        ppb = std::min(ppb, std::max(1, PM_LIG_LDS / (3 * in->n_lig) - 1));          // poses + the target in LDS
        ppb = std::min(ppb, std::max(1, PM_ROWS * 42 / n_perm));                    // per-automorphism results in LDS
      } else {
        ppb = std::min(ppb, std::max(1, PM_LIG_LDS / (3 * in->n_lig)));
      }
    } else {
      a.out.centroid = a.out.lig_rmsd = nullptr;
      ppb = std::min(ppb, std::max(1, PM_ROWS / in->n_res));
    }
    a.ppb = ppb;
    const long long blocks = (n_pf + ppb - 1) / ppb;
    if (blocks > 0x7fffffffLL) { dbfr_set_error("too many (pose, frame) pairs for one launch"); return DBFR_ERR_ARG; }
    hipLaunchKernelGGL(k_pose_metrics, dim3((unsigned)blocks), dim3(PM_THREADS), half == 0 ? PM_LIG_LDS * sizeof(float) : 0,
                       (hipStream_t)hip_stream, a);
    HIPCHECK(hipGetLastError());
  }
  return DBFR_OK;
}

// ------------------------------------------------------------------------------------------------ host: PDB text
#define RT_NAMES
#define RT_TABLE static const
namespace host_tables {
#include "residue_tables.inc"
}
#undef RT_TABLE
#undef RT_NAMES

namespace {

// druglib/utils/obj/protein.py:658-676: 1 -> A ... 26 -> Z, 27 -> AA, 28 -> BA, ...
std::string chain_tag(int chain_index) {
  long num = (long)chain_index;      // (chain_index + 1) - 1
  std::string s;
  while (num >= 0) {
    s.push_back((char)('A' + num % 26));
    num = num / 26 - 1;
  }
  return s;
}

struct Line {
  char buf[160];
  int n;
};

inline void pad80(std::string& out, const char* s, int n) {
  out.append(s, (size_t)n);
  if (n < 80) out.append((size_t)(80 - n), ' ');
  out.push_back('\n');
}

// The text of one structure.  Coordinates are float32 values printed through double ("%8.3f" of the exactly converted
// value, correctly rounded by glibc like Python's float formatting the reference relies on).
void format_structure(const dbfr_pdb_topology& t, const std::vector<int>& row_of, const float* pos14, int model, int add_end,
                      std::string& out) {
  out.clear();
  out.reserve((size_t)t.n_res * 9 * 81 + 256);
  char line[192];
  if (t.remark) pad80(out, t.remark, (int)strlen(t.remark));
  long serial = 1;
  const bool any_rows = pos14 != nullptr;
  for (int i = 0; i < t.n_res; ++i) {
    const int aa = t.aatype[i];
    const std::string tag = chain_tag(t.chain_index[i]);
    const char* rn = host_tables::kRestypeNames3[aa];
    const int ri = t.residue_index[i];
    const float* src14 = row_of[i] >= 0 ? pos14 + (size_t)row_of[i] * 42 : nullptr;
    for (int a = 0; a < 37; ++a) {
      if (t.atom37_mask[(size_t)i * 37 + a] < 0.5f) continue;
      const char* atn = host_tables::kAtom37Names[a];
      // Protein.pos_update with an atom14 array: atom37 slot a reads atom14 slot kAtom37ToAtom14[aa][a] (slot 0 for atoms
      // outside the residue type's atom14 set, e.g. OXT).  Residues the pose does not cover went through
      // Protein.to_pos14(True) first (export.py:262): masked by the type's atom14 set, so 'UNK' residues come out at 0.
      float via14[3];
      const float* p;
      if (src14) p = src14 + 3 * host_tables::kAtom37ToAtom14[aa][a];
      else if (!any_rows) p = t.atom37_pos + ((size_t)i * 37 + a) * 3;
      else {
        const int slot = host_tables::kAtom37ToAtom14[aa][a], a37 = host_tables::kAtom14ToAtom37[aa][slot];
        const float m = (float)host_tables::kAtom14Mask[aa][slot] * t.atom37_mask[(size_t)i * 37 + a37];
        for (int c = 0; c < 3; ++c) via14[c] = t.atom37_pos[((size_t)i * 37 + a37) * 3 + c] * m;
        p = via14;
      }
      char name[8];
      if (strlen(atn) == 4) snprintf(name, sizeof name, "%s", atn);
      else snprintf(name, sizeof name, " %s", atn);
      int n = snprintf(line, sizeof line, "ATOM  %5ld %-4s %3s %1s%4d    %8.3f%8.3f%8.3f%6.2f%6.2f          %2c  ", serial, name, rn,
                       tag.c_str(), ri, (double)p[0], (double)p[1], (double)p[2], 1.0, t.b_factors[(size_t)i * 37 + a], atn[0]);
      pad80(out, line, n);
      ++serial;
    }
    if (i == t.n_res - 1 || t.chain_index[i] != t.chain_index[i + 1]) {
      int n = snprintf(line, sizeof line, "TER   %5ld      %3s %1s%4d", serial, rn, tag.c_str(), ri);
      pad80(out, line, n);
      ++serial;
    }
  }
  if (model >= 0) pad80(out, "ENDMDL", 6);
  if (add_end) pad80(out, "END", 3);
}

int check_topology(const dbfr_pdb_topology* t, int n_rows, const int32_t* rows, const float* pos14, std::vector<int>& row_of) {
  if (!t || t->n_res < 0 || (t->n_res > 0 && (!t->aatype || !t->atom37_pos || !t->atom37_mask || !t->residue_index || !t->chain_index ||
                                              !t->b_factors))) {
    dbfr_set_error("null topology field");
    return DBFR_ERR_ARG;
  }
  for (int i = 0; i < t->n_res; ++i)
    if (t->aatype[i] < 0 || t->aatype[i] > 20) { dbfr_set_error("Invalid residue type found, only allow natural AA."); return DBFR_ERR_ARG; }
  row_of.assign((size_t)t->n_res, -1);
  if (n_rows < 0 || (n_rows > 0 && !pos14)) { dbfr_set_error("pos14 missing"); return DBFR_ERR_ARG; }
  if (n_rows > 0 && !rows && n_rows != t->n_res) { dbfr_set_error("rows == NULL needs n_rows == n_res"); return DBFR_ERR_ARG; }
  for (int k = 0; k < n_rows; ++k) {
    int r = rows ? rows[k] : k;
    if (r < 0 || r >= t->n_res) { dbfr_set_error("row index out of range"); return DBFR_ERR_ARG; }
    row_of[(size_t)r] = k;
  }
  return DBFR_OK;
}

}  // namespace

extern "C" int64_t dbfr_pdb_format(const dbfr_pdb_topology* topo, int32_t n_rows, const int32_t* rows, const float* pos14, int32_t model,
                                   int32_t add_end, char* out, int64_t cap) {
  try {                                   // no exception crosses the C ABI
    std::vector<int> row_of;
    int rc = check_topology(topo, n_rows, rows, pos14, row_of);
    if (rc) return rc;
    std::string s;
    format_structure(*topo, row_of, pos14, model, add_end, s);
    if ((int64_t)s.size() <= cap && out) memcpy(out, s.data(), s.size());
    return (int64_t)s.size();
  } catch (const std::exception& e) {
    dbfr_set_error(std::string("dbfr_pdb_format: ") + e.what());
    return DBFR_ERR_ARG;
  }
}

extern "C" int dbfr_pdb_write_files(const dbfr_pdb_topology* topo, int32_t n_rows, const int32_t* rows, const float* pos14, int32_t n_pose,
                                    const char* const* paths, int32_t n_threads) {
  try {                                   // no exception crosses the C ABI (thread creation / allocation may throw)
  std::vector<int> row_of;
  int rc = check_topology(topo, n_rows, rows, pos14, row_of);
  if (rc) return rc;
  if (n_pose < 0 || (n_pose > 0 && !paths)) { dbfr_set_error("paths missing"); return DBFR_ERR_ARG; }
  if (n_pose == 0) return DBFR_OK;
  int hw = (int)std::thread::hardware_concurrency();
  if (hw < 1) hw = 1;
  int nt = n_threads > 0 ? n_threads : std::min(hw, (int)n_pose);
  nt = std::max(1, std::min(nt, (int)n_pose));
  std::atomic<int> next(0), failed(-1);
  auto work = [&]() noexcept {
    try {
      std::string s;
      for (;;) {
        int i = next.fetch_add(1);
        if (i >= n_pose) break;
        format_structure(*topo, row_of, pos14 + (size_t)i * n_rows * 42, -1, 1, s);
        FILE* f = fopen(paths[i], "wb");
        bool ok = f && fwrite(s.data(), 1, s.size(), f) == s.size();
        if (f) ok = (fclose(f) == 0) && ok;
        if (!ok) { int e = -1; failed.compare_exchange_strong(e, i); }
      }
    } catch (...) { int e = -1; failed.compare_exchange_strong(e, 0); }
  };
  std::vector<std::thread> th;
  for (int k = 1; k < nt; ++k) {
    try { th.emplace_back(work); } catch (...) { break; }      // fewer helpers: the calling thread picks up the rest
  }
  work();
  for (auto& x : th) x.join();
  if (failed.load() >= 0) { dbfr_set_error(std::string("cannot write ") + paths[failed.load()]); return DBFR_ERR_ARG; }
  return DBFR_OK;
  } catch (const std::exception& e) {
    dbfr_set_error(std::string("dbfr_pdb_write_files: ") + e.what());
    return DBFR_ERR_ARG;
  }
}


// ------------------------------------------------------------------------------------------------ ligand SD files
// lig_final.sdf of every pose (DiffBindFR/evaluation/export.py:97-103,236-244: Ligand3D.pos_update + Chem.SDWriter): the
// ligand's V2000 mol block with the pose's coordinates in the atom lines.  Everything that is not a coordinate (header,
// the rest of each atom line, bond block, property block, data items) is text prepared once per ligand on the host
// (diffbindfr_amd/ligand.py: SdfTemplate); here only the 30 coordinate columns are formatted.  Host code, all pointers host.
static int check_sdf(const dbfr_sdf_template* t, const float* pos) {
  if (!t || !pos || t->n_atoms < 0 || !t->header || !t->trailer || (t->n_atoms > 0 && !t->atom_tail)) {
    dbfr_set_error("null argument (sdf template)");
    return DBFR_ERR_ARG;
  }
  for (int i = 0; i < t->n_atoms; ++i)
    if (!t->atom_tail[i]) { dbfr_set_error("null atom line (sdf template)"); return DBFR_ERR_ARG; }
  return DBFR_OK;
}

static void format_sdf(const dbfr_sdf_template& t, const float* pos, std::string& s) {
  s.clear();
  s += t.header;
  char buf[64];
  for (int i = 0; i < t.n_atoms; ++i) {
    // V2000 atom line: xxxxx.xxxxyyyyy.yyyyzzzzz.zzzz then " aaa dd ..." (CTfile spec; RDKit prints %10.4f as well)
    snprintf(buf, sizeof buf, "%10.4f%10.4f%10.4f", (double)pos[3 * i], (double)pos[3 * i + 1], (double)pos[3 * i + 2]);
    s += buf;
    s += t.atom_tail[i];
    s += '\n';
  }
  s += t.trailer;
}

extern "C" int64_t dbfr_sdf_format(const dbfr_sdf_template* t, const float* pos, char* out, int64_t cap) {
  try {
    if (check_sdf(t, pos)) return DBFR_ERR_ARG;
    std::string s;
    format_sdf(*t, pos, s);
    if (out && cap >= (int64_t)s.size()) memcpy(out, s.data(), s.size());
    return (int64_t)s.size();
  } catch (const std::exception& e) {
    dbfr_set_error(std::string("dbfr_sdf_format: ") + e.what());
    return DBFR_ERR_ARG;
  }
}

extern "C" int dbfr_sdf_write_files(const dbfr_sdf_template* t, const float* pos, int32_t n_pose, const char* const* paths,
                                    int32_t n_threads) {
  try {
    if (check_sdf(t, pos)) return DBFR_ERR_ARG;
    if (n_pose < 0 || (n_pose > 0 && !paths)) { dbfr_set_error("paths missing"); return DBFR_ERR_ARG; }
    if (n_pose == 0) return DBFR_OK;
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    int nt = n_threads > 0 ? n_threads : std::min(hw, (int)n_pose);
    nt = std::max(1, std::min(nt, (int)n_pose));
    std::atomic<int> next(0), failed(-1);
    auto work = [&]() noexcept {
      try {
        std::string s;
        for (;;) {
          int i = next.fetch_add(1);
          if (i >= n_pose) break;
          format_sdf(*t, pos + (size_t)i * t->n_atoms * 3, s);
          FILE* f = fopen(paths[i], "wb");
          bool ok = f && fwrite(s.data(), 1, s.size(), f) == s.size();
          if (f) ok = (fclose(f) == 0) && ok;
          if (!ok) { int e = -1; failed.compare_exchange_strong(e, i); }
        }
      } catch (...) { int e = -1; failed.compare_exchange_strong(e, 0); }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nt; ++k) {
      try { th.emplace_back(work); } catch (...) { break; }
    }
    work();
    for (auto& x : th) x.join();
    if (failed.load() >= 0) { dbfr_set_error(std::string("cannot write ") + paths[failed.load()]); return DBFR_ERR_ARG; }
    return DBFR_OK;
  } catch (const std::exception& e) {
    dbfr_set_error(std::string("dbfr_sdf_write_files: ") + e.what());
    return DBFR_ERR_ARG;
  }
}
