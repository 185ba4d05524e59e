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
// k_conv work decomposition (one workgroup = 256 threads = 4 waves = TE = 16 NB edges, NB = 3 by default):
//   phase A  gather edge_attr_ = [edge_emb | tab1[idx1][:48] | tab2[idx2][:48]] and x[gth] into LDS
//   phase B  h = relu(W1 a + b1): (K/16 row tiles) x (NB edge blocks) MFMA jobs split over the waves;
//            MFMA orientation D[row = weight row, col = edge]  => lane (g = lane>>4, n = lane&15)
//            owns ONE edge n for its whole life and 4 consecutive weight rows 4g..4g+3 per tile
//   phase C  every wave loads the full h^T as its MFMA B operand (K/4 VGPRs per 16-edge block)
//   phase D  every wave walks ITS OWN contiguous range of W2 row tiles.  Rows are pre-ordered at model
//            creation (api.cpp: pack_conv) so that a lane group (16 lanes = 16 edges x 4 rows) stays on ONE
//            output channel (io, w_out) for a run of tiles covering all paths and all u_in of that channel:
//            the lane contracts its 4 weights with x[u0..u0+3] (LDS) and the edge's harmonics through the
//            closed-form CG of the path, keeps the running message element in registers and stores it to the
//            LDS message tile exactly once -- no atomics, fixed summation order (bit-reproducible)
//   phase E  coalesced store of the [TE, D_out] message tile
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"

#ifndef CONV_PRIO
#define CONV_PRIO 0
#endif
#ifndef CONV_XPF
#define CONV_XPF 0
#endif
#ifndef CONV_NB2_WAVES
#define CONV_NB2_WAVES 2     // developer: 3 = three 32-edge workgroups per CU (needs <= 168 VGPRs)
#endif
#define XS_LD (MAXD + 4)   // 172: rows 16-B aligned (ds_read_b128), 16 consecutive rows hit 16 distinct 16-B slots

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TRACE_BLOCKS 1024
#define TRACE_TILES 48
#define TRACE_REC (2 + 3 * TRACE_TILES + 8)
#define RSTAMP(k) do { if ((ABL & 16) && a.trace && bid < TRACE_BLOCKS && lane == 0) a.trace[((size_t)bid * 4 + wave) * TRACE_REC + 2 + 3 * TRACE_TILES + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TSTAMP(k) do { if ((ABL & 16) && a.trace && bid < TRACE_BLOCKS && lane == 0) a.trace[((size_t)bid * 4 + wave) * TRACE_REC + 2 + 3 * TRACE_TILES + (k)] = __builtin_readcyclecounter(); } while (0)

// K: radial-MLP width (144 / 96); NB: 16-edge blocks per workgroup (TE = 16 NB edges share every A fragment
// fetched from L2 -- the L2->CU fabric, not HBM, is what the weight stream loads); ABL: developer ablations.
template <int K, int NB, int ABL>
__device__ __forceinline__ void conv_body(const ConvArgs& a, const int bid) {
  constexpr int TE = 16 * NB;
  constexpr int KT = K / 16;  // 16-wide tiles along the MLP input/hidden dim
  constexpr int KS = K / 4;   // MFMA k-steps
  constexpr int A_LD = K + 4;   // edge-major activation rows [edge][k], 16-B aligned: inputs a1, then the hidden layer h
  constexpr int UN = (TE * (MAXD + 8) > TE * A_LD) ? TE * (MAXD + 8) : TE * A_LD;
  __shared__ __attribute__((aligned(16))) float xs[TE * XS_LD];
  __shared__ __attribute__((aligned(16))) float un[UN];       // a1 [TE][A_LD]  ->  h [TE][A_LD]  ->  out [TE][O_LD]
  __shared__ float shs[TE * 10];
  __shared__ int s_idx[3 * TE];
  __shared__ __attribute__((aligned(16))) float ms[TE * 8];   // per edge: the symmetric traceless l=2 matrix (m00 m01 m02 m11 m12 m22) of PT_VTV

  const int E = min(*a.n_edges, a.max_edges);
  if (bid * TE >= E) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile0 = bid * TE;
  const int ne = min(TE, E - tile0);
  int* s_gth = s_idx;
  int* s_i1 = s_gth + TE;
  int* s_i2 = s_i1 + TE;
  const int n = lane & 15, g = lane >> 4;
  TSTAMP(0);
  const int D_in = a.w.D_in, D_out = a.w.D_out;
  const int O_LD = ((D_out + 3) & ~3) + 4;   // 16-B aligned rows (phase E moves float4s) + trash columns for padded channels

  if (tid < TE) {
    int e = tile0 + min(tid, ne - 1);
    s_gth[tid] = a.gth[e];
    s_i1[tid] = a.idx1[e];
    s_i2[tid] = (K == 144) ? a.idx2[e] : 0;
  }
  __syncthreads();
  // ---------------- phase A: gather edge_attr_ and x[gth] into LDS.  All 16-byte loads of a thread are issued
  // before the first LDS store so that one round of L2 latency covers the whole gather.
  {
    constexpr int NPART = (K == 144) ? 3 : 2;
    constexpr int P4 = NS / 4;                                   // float4 per 48-float part
    constexpr int NA4 = TE * NPART * P4, JA = (NA4 + 255) / 256;
    f32x4 va[JA];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int i = tid + 256 * j;
      va[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < NA4) {
        const int e = i / (NPART * P4), r = i - e * (NPART * P4), part = r / P4, c4 = r - part * P4;
        if (e < ne) {
          const float* src = part == 0 ? a.emb + (size_t)(tile0 + e) * NS
                           : part == 1 ? a.tab1 + (size_t)s_i1[e] * a.ld1 : a.tab2 + (size_t)s_i2[e] * a.ld2;
          va[j] = *reinterpret_cast<const f32x4*>(src + 4 * c4);
        }
      }
    }
    const int d4 = D_in >> 2, NX4 = TE * d4;
    constexpr int JX = (TE * (MAXD / 4) + 255) / 256;
    f32x4 vx[JX];
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int i = tid + 256 * j;
      vx[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < NX4) {
        const int e = i / d4, c4 = i - e * d4;
        if (e < ne) vx[j] = *reinterpret_cast<const f32x4*>(a.x + (size_t)s_gth[e] * a.ldx + 4 * c4);
      }
    }
    for (int i = tid; i < TE * SH_LD; i += 256) {
      int e = i / SH_LD, c = i - e * SH_LD;
      shs[e * 10 + c] = a.sh[(size_t)(tile0 + min(e, ne - 1)) * SH_LD + c];
    }
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int i = tid + 256 * j;
      if (i < NA4) {
        const int e = i / (NPART * P4), r = i - e * (NPART * P4);
        *reinterpret_cast<f32x4*>(un + e * A_LD + 4 * r) = va[j];
      }
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int i = tid + 256 * j;
      if (i < NX4) {
        const int e = i / d4, c4 = i - e * d4;
        *reinterpret_cast<f32x4*>(xs + e * XS_LD + 4 * c4) = vx[j];
      }
    }
  }
  __syncthreads();
  TSTAMP(1);
  if (tid < TE) {   // closed form of the 1 x 2 -> 1 coupling (must mirror so3_host.cpp: closed_form)
    const float r3 = 1.7320508075688772f;
    const float* sp = shs + tid * 10 + 4;
    const float s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4];
    float* m = ms + tid * 8;
    m[0] = -s2 - r3 * s4; m[1] = r3 * s1; m[2] = r3 * s0; m[3] = 2.f * s2; m[4] = r3 * s3; m[5] = -s2 + r3 * s4;
  }
  // ---------------- phase B: hidden layer on the matrix cores.  A job = one 16-row tile of W1 against all NB edge
  // blocks (each W1 fragment feeds NB MFMAs, NB independent accumulator chains); the next job's fragments are
  // requested before the current job's MFMAs.
  constexpr int JPW = (KT + 3) / 4;        // row tiles per wave
  f32x4 hacc[JPW][NB];
  {
    const f32x4* W1 = reinterpret_cast<const f32x4*>(a.w.W1p) + lane;
    f32x4 A1[KT];
    if (wave < KT) {
#pragma unroll
      for (int s4 = 0; s4 < KT; ++s4) A1[s4] = W1[((size_t)wave * KT + s4) * 64];
    }
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
      const int m = wave + 4 * jj;
#pragma unroll
      for (int b = 0; b < NB; ++b) hacc[jj][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (m < KT) {
        const int mn = (m + 4 < KT) ? m + 4 : m;
        const float* Bp = un + n * A_LD + 4 * g;     // k-step q of lane group g reads k = 16 s4 + 4 g + q (see api.cpp pack_conv)
#pragma unroll
        for (int s4 = 0; s4 < KT; ++s4) {
          const f32x4 av = A1[s4];
          f32x4 bq[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) bq[b] = *reinterpret_cast<const f32x4*>(Bp + 16 * b * A_LD + 16 * s4);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              hacc[jj][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bq[b][q], hacc[jj][b], 0, 0, 0);
          A1[s4] = W1[((size_t)mn * KT + s4) * 64];
        }
      }
    }
  }
  __syncthreads();  // everyone is done reading a1
#pragma unroll
  for (int jj = 0; jj < JPW; ++jj) {
    const int m = wave + 4 * jj;
    if (m < KT) {
      const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.w.b1 + 16 * m + 4 * g);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = fmaxf(hacc[jj][b][r] + b1v[r], 0.f);
        *reinterpret_cast<f32x4*>(un + (16 * b + n) * A_LD + 16 * m + 4 * g) = h;     // h[edge][hidden unit]
      }
    }
  }
  __syncthreads();
  // ---------------- phase C: B operand (h^T) into registers
  float Bv[NB][KS];
#pragma unroll
  for (int s4 = 0; s4 < KT; ++s4)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(un + (16 * b + n) * A_LD + 16 * s4 + 4 * g);
#pragma unroll
      for (int q = 0; q < 4; ++q) Bv[b][4 * s4 + q] = v[q];
    }
  __syncthreads();   // hs is dead: the region becomes the message tile (every element is stored exactly once)
  TSTAMP(2);
  RSTAMP(6);
  // ---------------- phase D: the W2 row tiles of this wave, run by run (channel-owner order, see api.cpp pack_conv).
  // Measured on MI355X (tools/exp/mfma_shadow.hip): a dense v_mfma_f32_16x16x4_f32 stream leaves room for only ~2.5
  // vector instructions of the OTHER wave on the SIMD per MFMA and for none of its own, so everything that is not an
  // MFMA is kept off the vector pipe: A fragments and biases come through buffer loads with scalar offsets (no address
  // VALU), the bias is the accumulator's initial value, the path type is dispatched once per run (tiles of a run
  // differ only in the u-quad), the harmonics are read once per run, and the contraction itself is 5..21 FMAs per
  // 16-edge block.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int r_begin = a.w.wave_run0[wv], r_end = a.w.wave_run0[wv + 1];
  const int t_last = a.w.wave_tile0[wv + 1] - 1;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w.W2p, 0, a.w.n_tiles * KT * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w.b2p, 0, a.w.n_tiles * 64, 0x00020000);
  const int vW = lane * 16, vB = g * 16;
  // A fragments of the current tile live in A[]; each register is re-loaded with the NEXT tile's fragment right after
  // its last MFMA, so a whole tile of matrix work hides the L2 latency with one register set.  Same for the bias.
  f32x4 A[KT], bias_n = {0.f, 0.f, 0.f, 0.f};
  if (r_begin < r_end) {
    const int t0 = a.w.wave_tile0[wv];
#pragma unroll
    for (int s4 = 0; s4 < KT; ++s4)
      A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, vW, (t0 * KT + s4) * 1024, 0));
    bias_n = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, vB, t0 * 64, 0));
  }
  float oacc[NB][3];   // running message element(s) of this lane's channel, per edge block
#pragma unroll
  for (int b = 0; b < NB; ++b) oacc[b][0] = oacc[b][1] = oacc[b][2] = 0.f;
  unsigned long long* trc = nullptr;
  int trc_n = 0;
  if ((ABL & 16) && a.trace && bid < TRACE_BLOCKS) {
    trc = a.trace + ((size_t)bid * 4 + wave) * TRACE_REC;
    if (lane == 0) { trc[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4); trc[1] = __builtin_amdgcn_s_getreg((3 << 11) | 20); }
  }
  const float* xs_lane = xs + n * XS_LD;      // + 16 b XS_LD per edge block (immediate offsets)
  const float* sh_lane = shs + n * 10;
  for (int r = ((ABL & 8) ? r_end : r_begin); r < r_end; ++r) {
    const RunDesc rd = a.w.runs[r];
    const int tile0 = rd.tile0_n & 0xfffff, nt = rd.tile0_n >> 20;
    const int type = rd.meta & 15, flags = (rd.meta >> 4) & 3, sh_off = (rd.meta >> 8) & 15, x_step = rd.meta >> 12;
    const int xo = (rd.x_off4 >> (8 * g)) & 0xff, oo = (rd.o_off4 >> (8 * g)) & 0xff;
    if (flags & 1) {
#pragma unroll
      for (int b = 0; b < NB; ++b) oacc[b][0] = oacc[b][1] = oacc[b][2] = 0.f;
    }
    auto run = [&](auto type_c) {
      constexpr int TYPE = decltype(type_c)::value;
      constexpr bool VIN = !(TYPE == PT_SS || TYPE == PT_SV);      // input irrep l=1: 12 consecutive floats per quad
      constexpr bool VOUT = !(TYPE == PT_SS || TYPE == PT_VVS);    // output irrep l=1
      constexpr int NSV = (TYPE == PT_SS || TYPE == PT_VS) ? 1 : (TYPE == PT_VTV ? 0 : 3);
      float S[NB][NSV ? NSV : 1];      // what the contraction needs of the edge's harmonics, constant over the run
      if (TYPE != PT_VTV) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float* sp = sh_lane + 160 * b + sh_off;
#pragma unroll
          for (int k = 0; k < NSV; ++k) S[b][k] = sp[k];
        }
      }
      const float* xp = xs_lane + xo;
      for (int i = 0; i < nt; ++i, xp += x_step) {
        const int t = tile0 + i;
        const int tn = t < t_last ? t + 1 : t;
        if ((ABL & 16) && trc && trc_n < TRACE_TILES && lane == 0) trc[2 + 3 * trc_n] = __builtin_readcyclecounter();
        f32x4 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = bias_n;
        bias_n = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, vB, tn * 64, 0));
#pragma unroll
        for (int s4 = 0; s4 < KT; ++s4) {
          const f32x4 av = A[s4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Bv[b][4 * s4 + q], acc[b], 0, 0, 0);
          // pin the re-load BEHIND the last MFMA that reads this register: hoisted to the top of the tile (what the
          // scheduler does on its own) it needs a second 36-register set and the kernel spills
          __builtin_amdgcn_sched_barrier(0);
          if (!(ABL & 2)) A[s4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, vW, (tn * KT + s4) * 1024, 0));
        }
        if ((ABL & 16) && trc && trc_n < TRACE_TILES) {
#pragma unroll
          for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(acc[b]));
          if (lane == 0) trc[3 + 3 * trc_n] = __builtin_readcyclecounter();
        }
        if (ABL & 1) {
#pragma unroll
          for (int b = 0; b < NB; ++b) asm volatile("" ::"v"(acc[b]));
          continue;
        }
#if CONV_PRIO == 1
        __builtin_amdgcn_s_setprio(3);
#elif CONV_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
#if CONV_XPF
        // x rows requested one edge block ahead: the LDS latency of block b+1 runs under the FMAs of block b
        f32x4 XA[NB], XB[NB], XC[NB];
        {
          const f32x4* x4 = reinterpret_cast<const f32x4*>(xp);
          XA[0] = x4[0];
          if (VIN) { XB[0] = x4[1]; XC[0] = x4[2]; }
        }
#endif
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const f32x4 v = acc[b];
#if CONV_XPF
          if (b + 1 < NB) {
            const f32x4* xn = reinterpret_cast<const f32x4*>(xp + 16 * (b + 1) * XS_LD);
            XA[b + 1] = xn[0];
            if (VIN) { XB[b + 1] = xn[1]; XC[b + 1] = xn[2]; }
          }
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 xa = XA[b];
#else
          const f32x4* x4 = reinterpret_cast<const f32x4*>(xp + 16 * b * XS_LD);
          const f32x4 xa = x4[0];
#endif
          if (!VIN) {
            const float z = v[0] * xa[0] + v[1] * xa[1] + v[2] * xa[2] + v[3] * xa[3];
            oacc[b][0] += z * S[b][0];
            if (TYPE == PT_SV) { oacc[b][1] += z * S[b][1]; oacc[b][2] += z * S[b][2]; }
          } else {
#if CONV_XPF
            const f32x4 xb = XB[b], xc = XC[b];
#else
            const f32x4 xb = x4[1], xc = x4[2];   // [u0..u0+3][3] = 12 consecutive floats
#endif
            const float z0 = v[0] * xa[0] + v[1] * xa[3] + v[2] * xb[2] + v[3] * xc[1];
            const float z1 = v[0] * xa[1] + v[1] * xb[0] + v[2] * xb[3] + v[3] * xc[2];
            const float z2 = v[0] * xa[2] + v[1] * xb[1] + v[2] * xc[0] + v[3] * xc[3];
            if (TYPE == PT_VS) {
              oacc[b][0] += z0 * S[b][0]; oacc[b][1] += z1 * S[b][0]; oacc[b][2] += z2 * S[b][0];
            } else if (TYPE == PT_VVS) {
              oacc[b][0] += z0 * S[b][0] + z1 * S[b][1] + z2 * S[b][2];
            } else if (TYPE == PT_VVV) {
              oacc[b][0] += z1 * S[b][2] - z2 * S[b][1];
              oacc[b][1] += z2 * S[b][0] - z0 * S[b][2];
              oacc[b][2] += z0 * S[b][1] - z1 * S[b][0];
            } else {   // PT_VTV: symmetric traceless matrix of the l=2 harmonics, read per tile (register budget)
              const f32x4 ma = *reinterpret_cast<const f32x4*>(ms + (16 * b + n) * 8);
              const float2 mb = *reinterpret_cast<const float2*>(ms + (16 * b + n) * 8 + 4);
              oacc[b][0] += ma[0] * z0 + ma[1] * z1 + ma[2] * z2;
              oacc[b][1] += ma[1] * z0 + ma[3] * z1 + mb.x * z2;
              oacc[b][2] += ma[2] * z0 + mb.x * z1 + mb.y * z2;
            }
            // one edge block at a time: the 12 x registers of the next block are not requested before this block's
            // FMAs are done (register budget 256 at 2 waves/SIMD; the other wave's MFMAs cover the LDS latency)
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#if CONV_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif CONV_PRIO == 2
        __builtin_amdgcn_s_setprio(3);
#endif
        if ((ABL & 16) && trc && trc_n < TRACE_TILES) {
#pragma unroll
          for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(oacc[b][0]));
          if (lane == 0) trc[4 + 3 * trc_n] = __builtin_readcyclecounter();
          ++trc_n;
        }
      }
      if (flags & 2) {   // last run of the channel group: this lane owns out[e][oo .. oo + (VOUT ? 3 : 1))
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float* op = un + (16 * b + n) * O_LD + oo;
          op[0] = oacc[b][0];
          if (VOUT) { op[1] = oacc[b][1]; op[2] = oacc[b][2]; }
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
  TSTAMP(3);
  RSTAMP(7);
  __syncthreads();
  TSTAMP(4);
  // ---------------- phase E
  {
    const int d4 = D_out >> 2;             // D_out is a multiple of 4 for every conv (84, 120, 168, 12, 96)
    for (int i = tid; i < ne * d4; i += 256) {
      const int e = i / d4, c4 = i - e * d4;
      *reinterpret_cast<f32x4*>(a.msg + (size_t)(tile0 + e) * D_out + 4 * c4) =
          *reinterpret_cast<const f32x4*>(un + e * O_LD + 4 * c4);
    }
  }
  TSTAMP(5);
}

template <int K, int NB, int ABL>
__global__ __launch_bounds__(256, (NB > 3 ? 1 : NB == 2 ? CONV_NB2_WAVES : 2)) void k_conv(ConvArgs a) {
  conv_body<K, NB, ABL>(a, blockIdx.x);
}

// The four convs of an interaction layer as ONE grid (bench-sized batches): blocks [first[c], first[c+1]) belong to conv c, so
// the partially filled last round of one conv is topped up by the first workgroups of the next instead of idling (4 tails
// per layer -> 1) and three launch gaps disappear.
struct ConvLayerArgs { ConvArgs c[4]; int first[5]; };
template <int NB>
__global__ __launch_bounds__(256, (NB > 3 ? 1 : NB == 2 ? CONV_NB2_WAVES : 2)) void k_conv_layer(ConvLayerArgs a) {
  const int b = blockIdx.x;
  const int c = b < a.first[1] ? 0 : b < a.first[2] ? 1 : b < a.first[3] ? 2 : 3;     // uniform: scalar loads of the conv's arguments
  conv_body<144, NB, 0>(a.c[c], b - a.first[c]);
}

#ifndef CONV_NB
#define CONV_NB 3
#endif

void launch_conv_layer(const ConvArgs* c4, hipStream_t st) {
  ConvLayerArgs a;
  const int te = 16 * CONV_NB;
  a.first[0] = 0;
  for (int i = 0; i < 4; ++i) { a.c[i] = c4[i]; a.c[i].trace = nullptr; a.first[i + 1] = a.first[i] + (c4[i].max_edges + te - 1) / te; }
  if (a.first[4] <= 0) return;
  hipLaunchKernelGGL((k_conv_layer<CONV_NB>), dim3(a.first[4]), dim3(256), 0, st, a);
}

void launch_conv(const ConvArgs& a, hipStream_t st) {
#ifdef DBFR_DEV_VARIANTS   // developer build (DBFR_BUILD_DEV=1 python -m diffbindfr_amd.build): timing-only variants, wrong results
  static int abl = -1;   // DBFR_CONV_ABL: 1 no contraction, 2 no W2 re-load, 8 prologue only
  if (abl < 0) { const char* e = getenv("DBFR_CONV_ABL"); abl = e ? atoi(e) : 0; }
#endif
  const int te = 16 * CONV_NB;
  const int tiles = (a.max_edges + te - 1) / te;
  if (tiles <= 0) return;
  ConvArgs b = a;
  b.trace = nullptr;
  static const char* trace_path = getenv("DBFR_CONV_TRACE");   // developer: dump a per-tile timeline of one big launch
  static int trace_skip = getenv("DBFR_CONV_TRACE_SKIP") ? atoi(getenv("DBFR_CONV_TRACE_SKIP")) : 0;   // matching launches to pass over first
  if (trace_path && a.w.K == 144 && a.w.W == 7776 && tiles >= 4096 && trace_skip-- <= 0) {
    const size_t nw = (size_t)TRACE_BLOCKS * 4 * TRACE_REC;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, nw * 8) == hipSuccess) {
      (void)hipMemsetAsync(d, 0, nw * 8, st);
      b.trace = d;
      hipLaunchKernelGGL((k_conv<144, CONV_NB, 16>), dim3(tiles), dim3(256), 0, st, b);
      (void)hipStreamSynchronize(st);
      std::vector<unsigned long long> h(nw);
      (void)hipMemcpy(h.data(), d, nw * 8, hipMemcpyDeviceToHost);
      (void)hipFree(d);
      if (FILE* f = fopen(trace_path, "wb")) { fwrite(h.data(), 8, nw, f); fclose(f); }
      trace_path = nullptr;
      return;
    }
  }
#define LAUNCH(KK, AB) hipLaunchKernelGGL((k_conv<KK, CONV_NB, AB>), dim3(tiles), dim3(256), 0, st, b)
  if (a.w.K != 144) { LAUNCH(96, 0); return; }
#ifdef DBFR_DEV_VARIANTS
  switch (abl) { case 1: LAUNCH(144, 1); return; case 2: LAUNCH(144, 2); return; case 8: LAUNCH(144, 8); return; default: break; }
#endif
  LAUNCH(144, 0);
#undef LAUNCH
}

// ------------------------------------------------------------------------------------------------
// Segmented mean + equivariant LayerNorm + residual (tpscore.py:190,196-197,513-516; :53-104).
// One wavefront per target node; the node's messages are contiguous rows [row_start, +row_cnt).
//   mode 0: out[n] = pad(old[n], D) + LN(mean)      mode 1: out[n] += LN(mean)     mode 2: out[n] = LN(mean)
// Sum of a node's message rows [rs, rs + rc), one float4 of columns per lane (lane < D / 4), in CSR order.  first != null (DBFR_GEMM_REDUCE_FIRST): the lanes
// of `sc_lanes` hold scalar-output columns, which k_convz writes in the FIRST row of every segment only (the other rows of these columns hold stale
// data): those lanes add flagged rows only.  The flags of up to 64 rows are fetched by one load and kept as a wave mask in scalar registers.
__device__ __forceinline__ f32x4 row_sum(const float* __restrict__ msg, int rs, int rc, int D, int lane, const uint8_t* __restrict__ first, unsigned long long sc_lanes) {
  const int d4 = D >> 2;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#ifdef RLN_NO_ROWS
  rc = min(rc, 1);
#endif
  const f32x4* r = reinterpret_cast<const f32x4*>(msg + (size_t)rs * D) + min(lane, d4 - 1);
  if (!first) {
    if (lane < d4) {
      int e = 0;
      // eight rows requested before the first add (the adds keep the CSR order: same bits as any other unrolling)
      for (; e + 8 <= rc; e += 8) {
        const f32x4 v0 = r[(size_t)e * d4], v1 = r[(size_t)(e + 1) * d4], v2 = r[(size_t)(e + 2) * d4], v3 = r[(size_t)(e + 3) * d4];
        const f32x4 v4 = r[(size_t)(e + 4) * d4], v5 = r[(size_t)(e + 5) * d4], v6 = r[(size_t)(e + 6) * d4], v7 = r[(size_t)(e + 7) * d4];
        acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
      }
      for (; e + 4 <= rc; e += 4) {
        const f32x4 v0 = r[(size_t)e * d4], v1 = r[(size_t)(e + 1) * d4], v2 = r[(size_t)(e + 2) * d4], v3 = r[(size_t)(e + 3) * d4];
        acc += v0; acc += v1; acc += v2; acc += v3;
      }
      for (; e < rc; ++e) acc += r[(size_t)e * d4];
    }
    return acc;
  }
  const bool sc = (sc_lanes >> lane) & 1ull;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int e0 = 0; e0 < rc; e0 += 64) {
    const int nb = min(64, rc - e0);
    const unsigned long long fl = __ballot(lane < nb && first[rs + e0 + lane] != 0);     // (all 64 lanes take part)
    if (lane < d4) {
      int e = 0;
      // eight (then four) rows requested before the first add; a row these columns were not written in contributes nothing
      auto row = [&](int i) { return (!sc || ((fl >> i) & 1ull)) ? r[(size_t)(e0 + i) * d4] : zero; };
      for (; e + 8 <= nb; e += 8) {
        const f32x4 v0 = row(e), v1 = row(e + 1), v2 = row(e + 2), v3 = row(e + 3), v4 = row(e + 4), v5 = row(e + 5), v6 = row(e + 6), v7 = row(e + 7);
        acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
      }
      for (; e + 4 <= nb; e += 4) {
        const f32x4 v0 = row(e), v1 = row(e + 1), v2 = row(e + 2), v3 = row(e + 3);
        acc += v0; acc += v1; acc += v2; acc += v3;
      }
      for (; e < nb; ++e)
        if (!sc || ((fl >> e) & 1ull)) acc += r[(size_t)(e0 + e) * d4];
    }
  }
  return acc;
}

__global__ __launch_bounds__(256) void k_reduce_ln(const float* __restrict__ msg, const int* __restrict__ row_start,
                                                   const int* __restrict__ row_cnt, int N, int D, LNDesc ln,
                                                   const float* __restrict__ old, int D_old, float* __restrict__ out,
                                                   int ldo, int mode, const uint8_t* first, unsigned long long sc_lanes) {
  __shared__ float buf[4][MAXD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int node_raw = blockIdx.x * 4 + wave;
  const bool live = node_raw < N;
  const int node = live ? node_raw : N - 1;
  const int rs = row_start[node], rc = row_cnt[node];
  // rows are 16-B aligned (D is a multiple of 4): one float4 per lane covers a row, four rows are requested before the
  // first add so that one L2 round trip serves four edges; the adds keep the edge order (reproducible, order = CSR order)
  const int d4 = D >> 2;
  const f32x4 acc = row_sum(msg, rs, rc, D, lane, first, sc_lanes);
  if (lane < d4) {
    const float cntf = (float)max(rc, 1);
    float* bw = buf[wave] + 4 * lane;
    bw[0] = acc[0] / cntf; bw[1] = acc[1] / cntf; bw[2] = acc[2] / cntf; bw[3] = acc[3] / cntf;
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
                      const float* old, int D_old, float* out, int ldo, int mode, hipStream_t st, const uint8_t* first, unsigned long long sc_lanes) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_reduce_ln, dim3((N + 3) / 4), dim3(256), 0, st, msg, row_start, row_cnt, N, D, ln, old, D_old,
                     out, ldo, mode, first, sc_lanes);
}

// ------------------------------------------------------------------------------------------------
// The four reductions of an interaction layer in ONE launch (used behind the fused k_conv2 layer launch, where every
// conv has its own message buffer): per ligand node  out = (pad(old) + LN_ll(mean ll)) + LN_al(mean al), per pocket atom
// out = (pad(old) + LN_aa(mean aa)) + LN_la(mean la) -- the very operation order of the mode 0 / mode 1 launch pair.
struct ReduceLayerArgs {
  const float* msg[4]; const int* row_start[4]; const int* row_cnt[4]; LNDesc ln[4];   // ll, al, aa, la
  int NL, NA, D, D_old;
  const float* old_l; const float* old_a; float* out_l; float* out_a;
  const uint8_t* first[4]; unsigned long long sc_lanes[4];   // DBFR_GEMM_REDUCE_FIRST: the segment-start flags of the four edge sets and the lanes of scalar-output columns (row_sum), else null / 0
};

__device__ __forceinline__ void mean_ln(const float* __restrict__ msg, int rs, int rc, int D, const LNDesc& ln, float* bw, int lane, const uint8_t* first, unsigned long long sc_lanes) {
  const int d4 = D >> 2;
  const f32x4 acc = row_sum(msg, rs, rc, D, lane, first, sc_lanes);
  if (lane < d4) {
    const float cntf = (float)max(rc, 1);
    bw[4 * lane] = acc[0] / cntf; bw[4 * lane + 1] = acc[1] / cntf; bw[4 * lane + 2] = acc[2] / cntf; bw[4 * lane + 3] = acc[3] / cntf;
  }
  __builtin_amdgcn_wave_barrier();
  int iw = 0, ib = 0;
#ifdef RLN_NO_LN
  return;
#endif
  for (int bk = 0; bk < ln.nblk; ++bk) {
    const int mul = ln.mul[bk], dim = ln.dim[bk], off = ln.off[bk];
    const int nel = mul * dim;
    float mean[3] = {0.f, 0.f, 0.f};
    const bool d3 = dim == 3;              // irreps are scalars or vectors: a constant divisor instead of a runtime division
    for (int i = lane; i < nel; i += 64) {
      int comp = d3 ? i % 3 : 0;
      float v = bw[off + i];
      if (comp == 0) mean[0] += v; else if (comp == 1) mean[1] += v; else mean[2] += v;
    }
    for (int o = 32; o > 0; o >>= 1) mean[0] += __shfl_xor(mean[0], o);
    mean[0] /= (float)mul;
    if (d3) {                              // scalar blocks have one component: two of the three butterflies are skipped (wave-uniform)
#pragma unroll
      for (int c = 1; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) mean[c] += __shfl_xor(mean[c], o);
        mean[c] /= (float)mul;
      }
    }
    float sq = 0.f;
    for (int i = lane; i < nel; i += 64) {
      int u = d3 ? i / 3 : i, comp = d3 ? i - 3 * u : 0;
      float v = bw[off + i] - (comp == 0 ? mean[0] : comp == 1 ? mean[1] : mean[2]) * ln.mean_shift[iw + u];
      bw[off + i] = v;
      sq += v * v;
    }
    __builtin_amdgcn_wave_barrier();
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float inv = 1.0f / sqrtf(sq / (float)nel + 1e-5f);
    for (int i = lane; i < nel; i += 64) {
      int u = d3 ? i / 3 : i;
      float v = bw[off + i] * (inv * ln.weight[iw + u]);
      if (ln.is0e[bk]) v += ln.bias[ib + u];
      bw[off + i] = v;
    }
    iw += mul;
    if (ln.is0e[bk]) ib += mul;
  }
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void k_reduce_ln_layer(ReduceLayerArgs a) {
  __shared__ float buf[4][2][MAXD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbl = (a.NL + 3) / 4;
  const bool lig = (int)blockIdx.x < nbl;
  const int node = (lig ? blockIdx.x : blockIdx.x - nbl) * 4 + wave;
  const int N = lig ? a.NL : a.NA;
  if (node >= N) return;                         // waves are independent here (wave-level barriers only)
  const int s0 = lig ? 0 : 2;
  mean_ln(a.msg[s0], a.row_start[s0][node], a.row_cnt[s0][node], a.D, a.ln[s0], buf[wave][0], lane, a.first[s0], a.sc_lanes[s0]);
  mean_ln(a.msg[s0 + 1], a.row_start[s0 + 1][node], a.row_cnt[s0 + 1][node], a.D, a.ln[s0 + 1], buf[wave][1], lane, a.first[s0 + 1], a.sc_lanes[s0 + 1]);
  const float* old = (lig ? a.old_l : a.old_a) + (size_t)node * a.D_old;
  float* out = (lig ? a.out_l : a.out_a) + (size_t)node * a.D;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = lane + 64 * j;
    if (c < a.D) out[c] = ((c < a.D_old ? old[c] : 0.f) + buf[wave][0][c]) + buf[wave][1][c];
  }
}

void launch_reduce_ln_layer(const ReduceLayerArgs& a, hipStream_t st) {
  const int nb = (a.NL + 3) / 4 + (a.NA + 3) / 4;
  if (nb <= 0) return;
  hipLaunchKernelGGL(k_reduce_ln_layer, dim3(nb), dim3(256), 0, st, a);
}
