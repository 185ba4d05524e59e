// Internal declarations shared by the host code and the HIP kernels of libdbfr.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/dbfr.h"

#define NS 48           // scalar multiplicity (cfg.ns), also the edge/node embedding width
#define NV 12           // vector multiplicity (cfg.nv)
#define SH_LD 9         // per-edge spherical-harmonic record (l=0..2), tor convs use 7 of 9
#define EMB 32          // sigma / distance embedding width
#define MAXD 168        // widest irreps feature: 48x0e+12x1o+12x1e+48x0o
#define EDGE_CHUNK 64   // targets per workgroup of the edge builder (graph.hip: k_edges), 4 lanes each
int dbfr_edge_chunks(int max_na, int max_nl);   // workgroups per (graph, edge set): sizes EdgeSet::g_cnt / g_base

// kernel-side tensor-product path types: closed forms of the real Wigner-3j tensors
enum PathType { PT_SS = 0, PT_SV = 1, PT_VS = 2, PT_VVS = 3, PT_VVV = 4, PT_VTV = 5 };

struct Irr { int mul, l, p; int dim() const { return 2 * l + 1; } };

struct PathDesc {
  int i1, i2, io, l1, l2, lo, mul1, mulo, w_off;
  float coeff;       // e3nn path_weight: sqrt((2lo+1)/sum fan-in)
  int type;          // PathType
  int in_off, sh_off, out_off;  // float offsets of the slots in x / sh / out rows
  float fold;        // coeff * closed-form CG scalar, folded into W2/b2 rows
};

struct ConvSpec {
  std::vector<Irr> in, sh, out;
  std::vector<PathDesc> paths;
  int K;             // n_edge_features = hidden width of the radial MLP (144 or 96)
  int W;             // tp.weight_numel
  int D_in, D_out;
};

// conv kinds: 0..3 layer convs by irreps depth, 4 final_conv, 5 tor / sc_tor convs
ConvSpec make_conv_spec(int kind);
void wigner3j_real(int l1, int l2, int l3, std::vector<double>& out);
int so3_selftest(std::string& err);

// ------------------------------------------------------------------ device-side descriptors
struct LNDesc {            // equivariant LayerNorm over the out irreps (tpscore.py:20-107)
  int nblk;
  int mul[4], dim[4], off[4], is0e[4];
  const float* mean_shift;   // [num_irreps]
  const float* weight;       // [num_irreps]
  const float* bias;         // [num 0e]
};

// A "run" = consecutive W2 row tiles of one wave that share the tensor-product path (same closed-form type, same
// harmonics) and walk the input multiplicity u in steps of 4: the kernel dispatches on the path type once per run and
// loads the edge harmonics once per run.  Uniform per run => fetched with scalar loads.
struct RunDesc {
  uint32_t tile0_n;   // first tile | n_tiles << 20
  uint32_t meta;      // type | flags << 4 (bit0: first run of its channel group, bit1: last) | sh_off << 8 | x_step << 12
                      // | (k_conv2 layouts) x phase << 20 | (k_conv2h) int8 k << 24: the run's W2h / b2h rows carry the factor 2^k
  uint32_t x_off4;    // byte g: x_off (floats) of lane group g at the run's first tile
  uint32_t o_off4;    // byte g: message column owned by lane group g
};

struct ConvW {             // one TensorProductConvLayer, device resident
  int K, D_in, D_out, n_tiles, W;
  int wave_tile0[5];  // tiles [wave_tile0[v], wave_tile0[v+1]) belong to wave v of the workgroup
  const float* W1p;   // [K/16][K/16][64][4]   lin.0 weight in MFMA A-fragment order
  const float* b1;    // [K]
  const float* W2p;   // [n_tiles][K/16][64][4] lin.3 weight rows permuted (path, w, u), path norm folded
  const float* b2p;   // [n_tiles*16]
  const RunDesc* runs;    // runs of all four waves, wave-major
  int wave_run0[5];       // runs [wave_run0[v], wave_run0[v+1]) belong to wave v
  LNDesc ln;
};

// Second-generation layout of the same conv for k_conv2 (conv2.hip): ONE tile sequence walked by every wave, channel
// groups of the outputs 0e / 1o first, then those of 1e / 0o (they read the 48x0o inputs through the LDS slot the 48x0e
// inputs occupied); contiguous group ranges = "parts" for the tail split.
struct ConvW2 {
  int D_in, D_out, n_tiles, n_runs;
  const float* W1p;   // shared with ConvW
  const float* b1;
  const float* W2q;   // [n_tiles][K/16][64][4]
  const float* b2q;   // [n_tiles*16]
  const void* W1h;    // [9] x 9280 B: lin.0 (W1p's tiles, bias rows) in the W2h tile format with the one factor 2^k1, for k_conv2h's hidden layer
  int k1;
  int f16_depth;      // largest row depth LEFT in the fp16 packing (api.cpp pack_f16_tiles): above F16_ROW_DEPTH_OK the conv is served by the fp32-instruction kernel k_conv2
  int f16_depth_run;  // ... before the per-row factors: above F16_ROW_DEPTH_OK the rows of W2h / W1h carry their own powers of two (W2rinv / W1rinv)
  const float* W2rinv;   // [n_tiles][16] 2^-d(row) of the rows of W2h packed with per-row factors, or null (every factor 1): k_conv2h<.., ROWF> takes them off the accumulator rows
  const float* W1rinv;   // [144] the same for W1h's rows (hidden units), or null
  const void* W2h;    // [n_tiles] x 9280 B: W2q x 2^k(run) cut into two fp16 pieces + the tile's 16 bias values x 2^k, for k_conv2h (conv2h.hip; k in RunDesc.meta bits 24..31)
  const RunDesc* runs;    // meta bit 20: run reads the second x layout; x offsets already mapped to the LDS row
  int part_run[4][9];     // part_run[si][p] = first run of part p when the conv is cut into 1 << si parts
};

struct Conv2Desc {         // one conv of a fused k_conv2 launch
  const int* n_edges; int max_edges;
  const int* gth;
  const float* emb;       // [E][NS]
  const float* sh;        // [E][SH_LD]
  const float* tab1; int ld1; const int* idx1;
  const float* tab2; int ld2; const int* idx2;
  const float* x; int ldx;
  ConvW2 w;
  float* msg;             // [E][D_out]
};

struct Conv2Args {
  Conv2Desc c[4];
  int n_conv;
  int* queue;             // [2] device ints, zero at launch: next unit, workgroups done (re-armed by the last one)
  int run_barrier, no_split;   // developer knobs (launch_conv2)
  int skew;                    // start delay of the second half of the waves, in 512-cycle sleeps
  unsigned long long* trace;   // developer read-out of k_conv2h (DBFR_CONV2_TRACE=<file>): [8 waves][C2_TRACE_CAP] shader-clock stamps of workgroup 0, or null
};
#define C2_TRACE_CAP 4096

// ---- reduce-first form of the SCALAR-OUTPUT paths of a K=144 conv (convz.hip, DBFR_GEMM_REDUCE_FIRST).
// A message element of an l = 0 output irrep is  sum_c y[e,c] (sum_k W2'[c,k,w] h[e,k] + b2'[c,w])  with c running over the (path, u_in)
// pairs into that irrep and y[e,c] = x[gth e, u] sh0[e] (scalar input) or xv[gth e, u] . sh1[e] (vector input): linear in y (x) h, and
// the scatter over the edges of a target node is linear too, so Z[t,c,k] = sum_{e -> t} y[e,c] h'[e,k] (h' = [h | 1]) is formed first
// and the big GEMM runs once per TARGET (segment of <= 32 edges), not once per edge.
#define CZ_NKT 10         // k tiles of 16: 144 hidden units + the constant 1 that carries the bias (tile 9, column 0)
#define CZ_MAXCT 12       // c tiles of 16 over both scalar output irreps (layer convs: 8, torsion convs: 10)
#define CZ_MAXSEG 4       // targets (segments) per chunk: k_convz keeps the masked y operand of every segment in registers
#define CZ_TILE_BYTES 6144   // W2' of one (c tile, k tile, k-step): [3 w tiles][hi, lo][64 lanes][8 fp16]
struct ConvZ {
  int n_io;               // scalar output irreps of this conv (1 or 2)
  int nct[2];             // c tiles per irrep
  int ct0[2];             // first c tile of the irrep in cdesc / W2z
  int out_off[2];         // message column of the irrep's channel 0 (48 channels each)
  int nc_valid[2];        // (path, u_in) pairs into the irrep without the padding of its last c tile (profiling: the flops that are not padding)
  const uint32_t* cdesc;  // [n c tiles][16]: x offset (floats) | l_in << 12 | sh offset << 16 | valid << 31
  const void* W2z;        // [c tile][k tile][8 k-steps][3 w tiles][hi, lo][64 lanes][8 fp16] x 2^s(w), s per output row
  const float* rowinv;    // [n_io][48]: 2^-s(w)
  const void* W1h; int k1;   // lin.0 as in ConvW2
  const float* W1rinv;    // [144] per-row factors of W1h to take off the hidden units, or null (ConvW2)
};
struct ConvZDesc {
  const int* n_edges; int max_edges;
  const int* tgt;         // scatter target of every edge (edges sorted by target)
  const int* gth;
  const float* emb; const float* sh;
  const float* tab1; int ld1; const int* idx1;
  const float* tab2; int ld2; const int* idx2;
  const float* x; int ldx;
  ConvZ w;
  float* msg; int D_out;
  const int* chunk_es;    // [n_chunks] first edge of every chunk (k_chunk_fill: <= 32 edges AND <= CZ_MAXSEG targets of ONE graph: batch-independent sums)
  const int* chunk_gl;    // [n_chunks] graph << 6 | number of edges
  const int* n_chunks;    // device scalar
  int max_chunks;
  const float* xmax;      // [graphs] largest |x| over the rows of every graph (k_row_absmax), or null: the kernel reads the chunk's gathered rows itself
};
struct ConvZArgs { ConvZDesc c[4]; int n_conv; float* dbg; double* executed; int dbg_sel; };   // executed (profiling only): [0] += flops of the matrix instructions issued, [1] += the flops among them that are not padding (live edge slots, live segment columns, valid (path, u) pairs)   // dbg (developer, DBFR_CONVZ_DEBUG=<file>): workgroup 0 / wave 0 of the first unit dumps h [32 slots][144]

static inline uint16_t dbfr_bf16_rne(float x) {   // round-to-nearest-even fp32 -> bf16 (finite inputs)
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float dbfr_bf16_to_f32(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct Mlp2 {              // SimpleLinear: Linear(in,hid) -> act -> Linear(hid,out)
  int in, hid, out;
  const float* w0t;   // [in][hid]   (transposed for coalesced/bank-friendly reads)
  const float* b0;    // [hid] or null
  const float* w1t;   // [hid][out]
  const float* b1;    // [out] or null
};

struct EdgeSet {           // one per-step edge list, grouped (CSR) by scatter-target node
  int cap;
  int* n_edges;        // device scalar
  int* tgt;            // [cap] scatter target (reference edge_index[0] of the conv call)
  int* gth;            // [cap] gather source  (reference edge_index[1])
  int* aux;            // [cap] bond index for ligand bond edges, -1 otherwise
  float* dist;         // [cap] |edge_vec|
  float* sh;           // [cap][SH_LD]
  float* emb;          // [cap][NS] edge embedding after its SimpleLinear
  int* row_start;      // [n_targets]
  int* row_cnt;        // [n_targets]
  int* g_cnt;          // [G * n_chunk] per-(graph, target chunk) totals (count pass); n_chunk = dbfr_edge_chunks()
  int* g_base;         // [G * n_chunk] base offset of each chunk's first edge (scan)
  int* chunk0;         // [G + 1] first chunk of every graph (k_chunk_count -> k_graph_chunks; chunks never straddle graphs: convz.hip)
  int* gedge0;         // [G + 1] first edge of every graph
  int* chunk_es;       // [chunk_cap] first edge of every chunk (k_chunk_fill)
  int* chunk_gl;       // [chunk_cap] graph << 6 | number of edges (<= 32, holding <= CZ_MAXSEG targets)
  int chunk_cap;
  uint8_t* seg_first;  // [cap] 1: the edge starts a segment (first edge of its target inside its chunk) -- k_convz stores a segment's scalar-output sum in that
                       // message row and writes nothing into the scalar columns of the other rows; the reductions read those columns of flagged rows only
};

struct ConvArgs {
  const int* n_edges;
  int max_edges;
  const int* tgt;
  const int* gth;
  const float* emb;       // [E][NS]
  const float* sh;        // [E][SH_LD]
  float sh_sign;          // -1 flips the l=1 components (cross_la reuses the al harmonics)
  // radial-MLP input = [emb | tab1[idx1[e]][:NS] | tab2[idx2[e]][:NS]] (K=144) or [emb | tab1..] (K=96)
  const float* tab1; int ld1; const int* idx1;
  const float* tab2; int ld2; const int* idx2;
  const float* x; int ldx;      // tensor-product input rows, gathered by gth
  ConvW w;
  float* msg;                   // [E][D_out]
  unsigned long long* trace;    // developer timeline (DBFR_CONV_TRACE): [block][wave][1 + 3*TRACE_TILES] or null
};

#define HIPCHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      dbfr_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                       \
      return DBFR_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

void dbfr_set_error(const std::string& s);
// Compute units of the CURRENT device (a process may drive several devices; cached per device id).
int dbfr_current_cu_count();
// A kernel launcher could not prepare its launch (e.g. hipFuncSetAttribute refused the LDS size): recorded on this thread and returned
// as DBFR_ERR_HIP by the entry point that issued the launch (api.cpp: take_launch_error).  Returns true when `e` is an error.
bool dbfr_launch_check(hipError_t e, const char* what);
