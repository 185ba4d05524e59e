// Fused tensor-product convolution for gfx950 (MI355X / CDNA4).
//
// Replaces, per TensorProductConvLayer call (tpscore.py:177-199):
//     w   = fc(edge_attr_)                      SimpleLinear 144->144->W   (torch GEMMs)
//     m   = FullyConnectedTensorProduct(x[dst], sh, w)                     (e3nn codegen)
//     out = scatter(m, src, reduce='mean') ; LayerNorm(out)                (torch_scatter)
// by two kernels:
//   k_conv<K>      radial MLP on the matrix cores (exact-fp32 v_mfma_f32_16x16x4_f32) with the
//                  per-edge weights consumed in registers by the Clebsch-Gordan contraction --
//                  the [E, W] weight tensor (31 KB/edge at W=7776) never exists in HBM;
//                  emits the per-edge message [E, D_out].
//   k_reduce_ln    wavefront segmented mean over the CSR-grouped messages of each target node
//                  (fixed order => reproducible), equivariant LayerNorm, residual accumulate.
//
// k_conv work decomposition (one workgroup = 256 threads = 4 waves = TE=32 edges):
//   phase A  gather edge_attr_ = [edge_emb | tab1[idx1][:48] | tab2[idx2][:48]] and x[gth] into LDS
//   phase B  h = relu(W1 a + b1): (K/16 row tiles) x (2 edge blocks) MFMA jobs split over the waves;
//            MFMA orientation D[row = weight row, col = edge]  => lane (g = lane>>4, n = lane&15)
//            owns ONE edge n for its whole life and 4 consecutive weight rows 4g..4g+3 per tile
//   phase C  every wave loads the full h^T as its MFMA B operand (K/4 VGPRs per 16-edge block)
//   phase D  wave w walks W2 row tiles t = w, w+4, ...; rows are pre-permuted (path, w_out, u_in) with
//            U in {12,48} so every 4-row quad shares one output channel: the lane contracts its 4
//            weights with x[u0..u0+3] (LDS) and the edge's harmonics through the closed-form CG of the
//            path and adds the result into the LDS message tile with ds_add_f32
//   phase E  coalesced store of the [TE, D_out] message tile
#include "common.h"

#define TE 32
#define XS_LD (MAXD + 1)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(256) void k_conv(ConvArgs a) {
  constexpr int KT = K / 16;  // 16-wide tiles along the MLP input/hidden dim
  constexpr int KS = K / 4;   // MFMA k-steps
  constexpr int A_LD = K + 1;
  constexpr int H_LD = TE + 1;
  constexpr int UN = (TE * XS_LD > K * H_LD) ? TE * XS_LD : K * H_LD;
  __shared__ float xs[TE * XS_LD];
  __shared__ float un[UN];       // a1 [TE][A_LD]  ->  hs [K][H_LD]  ->  out [TE][D_out+1]
  __shared__ float shs[TE * 10];
  __shared__ int s_gth[TE], s_i1[TE], s_i2[TE];

  const int E = min(*a.n_edges, a.max_edges);
  const int tile0 = blockIdx.x * TE;
  if (tile0 >= E) return;
  const int ne = min(TE, E - tile0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int D_in = a.w.D_in, D_out = a.w.D_out;
  const int O_LD = D_out + 1;

  if (tid < TE) {
    int e = tile0 + min(tid, ne - 1);
    s_gth[tid] = a.gth[e];
    s_i1[tid] = a.idx1[e];
    s_i2[tid] = (K == 144) ? a.idx2[e] : 0;
  }
  __syncthreads();
  // ---------------- phase A
  for (int i = tid; i < TE * NS; i += 256) {
    int e = i / NS, c = i - e * NS;
    int ge = tile0 + min(e, ne - 1);
    float live = e < ne ? 1.f : 0.f;
    un[e * A_LD + c] = live * a.emb[(size_t)ge * NS + c];
    un[e * A_LD + NS + c] = live * a.tab1[(size_t)s_i1[e] * a.ld1 + c];
    if (K == 144) un[e * A_LD + 2 * NS + c] = live * a.tab2[(size_t)s_i2[e] * a.ld2 + c];
  }
  for (int i = tid; i < TE * D_in; i += 256) {
    int e = i / D_in, c = i - e * D_in;
    xs[e * XS_LD + c] = e < ne ? a.x[(size_t)s_gth[e] * a.ldx + c] : 0.f;
  }
  for (int i = tid; i < TE * SH_LD; i += 256) {
    int e = i / SH_LD, c = i - e * SH_LD;
    float v = a.sh[(size_t)(tile0 + min(e, ne - 1)) * SH_LD + c];
    if (c >= 1 && c <= 3) v *= a.sh_sign;
    shs[e * 10 + c] = v;
  }
  __syncthreads();
  // ---------------- phase B: hidden layer on the matrix cores
  constexpr int NJOB = KT * 2;
  constexpr int JPW = (NJOB + 3) / 4;
  f32x4 hacc[JPW];
#pragma unroll
  for (int jj = 0; jj < JPW; ++jj) {
    int job = wave + 4 * jj;
    hacc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (job < NJOB) {
      int m = job >> 1, b = job & 1;
      const f32x4* Ap = reinterpret_cast<const f32x4*>(a.w.W1p) + (size_t)m * KT * 64 + lane;
      const float* Bp = un + (16 * b + n) * A_LD + g;
#pragma unroll
      for (int s4 = 0; s4 < KT; ++s4) {
        f32x4 A4 = Ap[s4 * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          hacc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(A4[q], Bp[4 * (4 * s4 + q)], hacc[jj], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // everyone is done reading a1
#pragma unroll
  for (int jj = 0; jj < JPW; ++jj) {
    int job = wave + 4 * jj;
    if (job < NJOB) {
      int m = job >> 1, b = job & 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 16 * m + 4 * g + r;
        un[row * H_LD + 16 * b + n] = fmaxf(hacc[jj][r] + a.w.b1[row], 0.f);
      }
    }
  }
  __syncthreads();
  // ---------------- phase C: B operand (h^T) into registers
  float B0[KS], B1[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    B0[s] = un[(4 * s + g) * H_LD + n];
    B1[s] = un[(4 * s + g) * H_LD + 16 + n];
  }
  __syncthreads();
  for (int i = tid; i < TE * O_LD; i += 256) un[i] = 0.f;
  __syncthreads();
  // ---------------- phase D: W2 row tiles
  const f32x4* W2 = reinterpret_cast<const f32x4*>(a.w.W2p);
  for (int t = wave; t < a.w.n_tiles; t += 4) {
    const f32x4* Ap = W2 + (size_t)t * KT * 64 + lane;
    f32x4 A[KT];
#pragma unroll
    for (int s4 = 0; s4 < KT; ++s4) A[s4] = Ap[s4 * 64];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < KT; ++s4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s4][q], B0[4 * s4 + q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s4][q], B1[4 * s4 + q], acc1, 0, 0, 0);
      }
    }
    const uint32_t qd = a.w.quads[t * 4 + g];
    const int x_off = qd & 0xff, o_off = (qd >> 8) & 0xff, sh_off = (qd >> 20) & 0xf;
    const int type = __builtin_amdgcn_readfirstlane((qd >> 16) & 0xf);  // uniform per tile
    const f32x4 bias = reinterpret_cast<const f32x4*>(a.w.b2p)[t * 4 + g];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int e = 16 * b + n;
      f32x4 v = (b == 0 ? acc0 : acc1) + bias;
      const float* xr = xs + e * XS_LD + x_off;
      const float* sp = shs + e * 10 + sh_off;
      float* op = un + e * O_LD + o_off;
      if (type == PT_SS || type == PT_SV) {
        float z = v[0] * xr[0] + v[1] * xr[1] + v[2] * xr[2] + v[3] * xr[3];
        if (type == PT_SS) {
          atomicAdd(op, z * sp[0]);
        } else {
          atomicAdd(op + 0, z * sp[0]);
          atomicAdd(op + 1, z * sp[1]);
          atomicAdd(op + 2, z * sp[2]);
        }
      } else {
        float z0 = v[0] * xr[0] + v[1] * xr[3] + v[2] * xr[6] + v[3] * xr[9];
        float z1 = v[0] * xr[1] + v[1] * xr[4] + v[2] * xr[7] + v[3] * xr[10];
        float z2 = v[0] * xr[2] + v[1] * xr[5] + v[2] * xr[8] + v[3] * xr[11];
        if (type == PT_VS) {
          float s0 = sp[0];
          atomicAdd(op + 0, z0 * s0);
          atomicAdd(op + 1, z1 * s0);
          atomicAdd(op + 2, z2 * s0);
        } else if (type == PT_VVS) {
          atomicAdd(op, z0 * sp[0] + z1 * sp[1] + z2 * sp[2]);
        } else if (type == PT_VVV) {
          float s0 = sp[0], s1 = sp[1], s2 = sp[2];
          atomicAdd(op + 0, z1 * s2 - z2 * s1);
          atomicAdd(op + 1, z2 * s0 - z0 * s2);
          atomicAdd(op + 2, z0 * s1 - z1 * s0);
        } else {  // PT_VTV
          const float r3 = 1.7320508075688772f;
          float s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4];
          float m00 = -s2 - r3 * s4, m01 = r3 * s1, m02 = r3 * s0, m11 = 2.f * s2, m12 = r3 * s3,
                m22 = -s2 + r3 * s4;
          atomicAdd(op + 0, m00 * z0 + m01 * z1 + m02 * z2);
          atomicAdd(op + 1, m01 * z0 + m11 * z1 + m12 * z2);
          atomicAdd(op + 2, m02 * z0 + m12 * z1 + m22 * z2);
        }
      }
    }
  }
  __syncthreads();
  // ---------------- phase E
  for (int i = tid; i < ne * D_out; i += 256) {
    int e = i / D_out, c = i - e * D_out;
    a.msg[(size_t)(tile0 + e) * D_out + c] = un[e * O_LD + c];
  }
}

void launch_conv(const ConvArgs& a, hipStream_t st) {
  int blocks = (a.max_edges + TE - 1) / TE;
  if (blocks <= 0) return;
  if (a.w.K == 144)
    hipLaunchKernelGGL(k_conv<144>, dim3(blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_conv<96>, dim3(blocks), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------------------------------------
// Segmented mean + equivariant LayerNorm + residual (tpscore.py:190,196-197,513-516; :53-104).
// One wavefront per target node; the node's messages are contiguous rows [row_start, +row_cnt).
//   mode 0: out[n] = pad(old[n], D) + LN(mean)      mode 1: out[n] += LN(mean)     mode 2: out[n] = LN(mean)
__global__ __launch_bounds__(256) void k_reduce_ln(const float* __restrict__ msg, const int* __restrict__ row_start,
                                                   const int* __restrict__ row_cnt, int N, int D, LNDesc ln,
                                                   const float* __restrict__ old, int D_old, float* __restrict__ out,
                                                   int ldo, int mode) {
  __shared__ float buf[4][MAXD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int node_raw = blockIdx.x * 4 + wave;
  const bool live = node_raw < N;
  const int node = live ? node_raw : N - 1;
  const int rs = row_start[node], rc = row_cnt[node];
  float acc[3] = {0.f, 0.f, 0.f};
  for (int e = 0; e < rc; ++e) {
    const float* r = msg + (size_t)(rs + e) * D;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int c = lane + 64 * j;
      if (c < D) acc[j] += r[c];
    }
  }
  const float cntf = (float)max(rc, 1);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int c = lane + 64 * j;
    if (c < D) buf[wave][c] = acc[j] / cntf;
  }
  __syncthreads();
  int iw = 0, ib = 0;
  for (int bk = 0; bk < ln.nblk; ++bk) {
    const int mul = ln.mul[bk], dim = ln.dim[bk], off = ln.off[bk];
    const int nel = mul * dim;
    // mean over multiplicity per component, learnable shift
    float mean[3] = {0.f, 0.f, 0.f};
    for (int i = lane; i < nel; i += 64) {
      int comp = i % dim;
      float v = buf[wave][off + i];
      if (comp == 0) mean[0] += v; else if (comp == 1) mean[1] += v; else mean[2] += v;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      for (int o = 32; o > 0; o >>= 1) mean[c] += __shfl_xor(mean[c], o);
      mean[c] /= (float)mul;
    }
    float sq = 0.f;
    for (int i = lane; i < nel; i += 64) {
      int u = i / dim, comp = i - u * dim;
      float v = buf[wave][off + i] - (comp == 0 ? mean[0] : comp == 1 ? mean[1] : mean[2]) * ln.mean_shift[iw + u];
      buf[wave][off + i] = v;
      sq += v * v;
    }
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float inv = 1.0f / sqrtf(sq / (float)nel + 1e-5f);
    for (int i = lane; i < nel; i += 64) {
      int u = i / dim;
      float v = buf[wave][off + i] * (inv * ln.weight[iw + u]);
      if (ln.is0e[bk]) v += ln.bias[ib + u];
      buf[wave][off + i] = v;
    }
    iw += mul;
    if (ln.is0e[bk]) ib += mul;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int c = lane + 64 * j;
    if (c < D && live) {
      float v = buf[wave][c];
      float* o = out + (size_t)node * ldo + c;
      if (mode == 0) v = (c < D_old ? old[(size_t)node * D_old + c] : 0.f) + v;
      else if (mode == 1) v = *o + v;
      *o = v;
    }
  }
}

void launch_reduce_ln(const float* msg, const int* row_start, const int* row_cnt, int N, int D, const LNDesc& ln,
                      const float* old, int D_old, float* out, int ldo, int mode, hipStream_t st) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_reduce_ln, dim3((N + 3) / 4), dim3(256), 0, st, msg, row_start, row_cnt, N, D, ln, old, D_old,
                     out, ldo, mode);
}
