// libdbfr host side: model packing, workspace planning, kernel orchestration, C ABI (include/dbfr.h).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>

#include "common.h"

// ---- kernel launchers (conv.hip / graph.hip / heads.hip)
void launch_conv(const ConvArgs& a, hipStream_t st);
void launch_conv_layer(const ConvArgs* c4, hipStream_t st);
void launch_conv2(const Conv2Args& a, hipStream_t st);
void launch_conv2h(const Conv2Args& a, hipStream_t st);
void launch_convz(const ConvZArgs& a, hipStream_t st);
void launch_reduce_ln(const float* msg, const int* row_start, const int* row_cnt, int N, int D, const LNDesc& ln,
                      const float* old, int D_old, float* out, int ldo, int mode, hipStream_t st, const uint8_t* first = nullptr, unsigned long long sc_lanes = 0);
struct ReduceLayerArgs {
  const float* msg[4]; const int* row_start[4]; const int* row_cnt[4]; LNDesc ln[4];
  int NL, NA, D, D_old;
  const float* old_l; const float* old_a; float* out_l; float* out_a;
  const uint8_t* first[4]; unsigned long long sc_lanes[4];   // (conv.hip: DBFR_GEMM_REDUCE_FIRST reads the scalar-output columns of a segment's first row only)
};
// the lanes (one float4 of message columns each) that hold the scalar-output columns k_convz writes: 12 per irrep from out_off / 4
static unsigned long long convz_sc_lanes(const ConvZ& z) {
  unsigned long long m = 0;
  for (int i = 0; i < z.n_io; ++i) m |= 0xfffull << (z.out_off[i] / 4);
  return m;
}
void launch_reduce_ln_layer(const ReduceLayerArgs& a, hipStream_t st);
enum SetKind { SET_LL = 0, SET_AA = 1, SET_AL = 2, SET_LA = 3, SET_TOR = 4, SET_SC = 5, N_SETS = 6 };
struct GraphArgs {
  dbfr_batch b;
  const int* lig_batch; const int* atm_batch; const uint8_t* is_cab; const int* n_cab; const float* tr_sigma;
  float lig_cut2, atom_cut2, cross_cut2;
  int lig_cap, atom_cap, dynamic_cross;
  EdgeSet set[N_SETS];
  int* err;
  int step, lds_nl, lds_na, n_chunk, lanes;
};
void dbfr_edge_form(const dbfr_batch& b, int* n_chunk, int* lanes);
void launch_edges(const GraphArgs& A, bool heads_only, hipStream_t st);
void launch_edge_log(const GraphArgs& A, int* log_row, int stride, hipStream_t st);
void launch_edge_ties(const GraphArgs& A, int* log_row, int stride, float tol, hipStream_t st);
void launch_graph_chunks(const GraphArgs& A, hipStream_t st);
void launch_flat_chunks(const int* tgt, const int* n_edges, int max_edges, int span, int n_span, int* cnt0, int cap, int* chunk_es, int* chunk_gl, hipStream_t st);
void launch_row_absmax(const float* lx, const int* lig_ptr, float* out_l, const float* ax, const int* atm_ptr, float* out_a, int ld, int D, int G, int n_rows_a, hipStream_t st);
void launch_batch_vectors(const dbfr_batch& b, int* lig_batch, int* atm_batch, uint8_t* is_cab, int* n_cab,
                          int* tor_batch, int* sc_batch, hipStream_t st);
void launch_time_embed(const float* t, int G, float emb_scale, float* temb, hipStream_t st);
enum MlpIn { IN_LIGNODE = 0, IN_LIGEDGE = 1, IN_TG = 2, IN_G = 3 };
struct MlpArgs {
  Mlp2 w; int mode; const int* n_rows_dev; int n_rows_max; const float* temb; const int* row_graph_tab; const int* tgt;
  const int* aux; const float* dist; const float* bond_feat; int nfeat; const float* lig_node; int nnode;
  const float* gs_offset; const float* gs_coeff; float* out;
};
void launch_mlp(const MlpArgs& a, hipStream_t st);
struct AtomEncArgs {
  const float* pocket_feat; const int* atm_batch; const float* temb; const float* emb[5]; int dims[5];
  const float* lin_t; int NA; float* out;
};
void launch_atom_encoder(const AtomEncArgs& a, hipStream_t st);
void launch_center_edges(const dbfr_batch& b, int* tgt, int* gth, float* dist, float* sh, int* row_start, int* row_cnt,
                         hipStream_t st);
struct TrRotArgs {
  const float* gp; const float* temb; const float* tr_sigma; const float* rot_norm; Mlp2 tr, rot; int G;
  int scale_by_sigma; float* tr_out; float* rot_out; int* err;
};
void launch_trrot(const TrRotArgs& a, hipStream_t st);
void launch_bond_attr(const float* x, int ldx, const int* b0, const int* b1, const int* bsel, int stride, int n,
                      float* out, hipStream_t st);
void launch_tor_final(const float* feat, const Mlp2& w, const float* norm2, int scale, int n, float* out, hipStream_t st);
struct SdeLigArgs {
  dbfr_batch b; const float* tr_score; const float* rot_score; const float* tor_score; const float* z_tr;
  const float* z_rot; const float* z_tor; float dt, tr_g2, tr_gsdt, rot_g2, rot_gsdt, tor_g2, tor_gsdt; float* traj;
  int* err;
};
void launch_sde_ligand(const SdeLigArgs& a, hipStream_t st);
void launch_sidechain(const dbfr_batch& b, const float* score, const float* z, float dt, float g2, float gsdt,
                      const int* a14_group, float* atom14_out, float* traj14, const int* err, hipStream_t st);
void launch_fill(float* p, float v, int n, hipStream_t st);
void launch_set_int(int* p, int v, hipStream_t st);
void launch_init_poses(const dbfr_batch& b, const dbfr_init_tape& z, const int* a14_group, float* atom14_out, hipStream_t st);
void launch_extract_templates(int n_res, const int* aatype, const float* pos14, float* transl, float* rots, float* frames,
                              float* rigid, float* angle, hipStream_t st);
void launch_select_pocket(int n_prot, int n_res_total, const int* res_ptr, int m, const float* pos, const float* mask, const int* ref_ptr,
                          const float* ref, float cut2, int max_neighbors, float* d2, unsigned char* out, hipStream_t st);
void launch_acc_flops(const int* n_edges, double flops_per_edge, double bytes_per_edge, double fused_bytes_per_edge, double* counter, hipStream_t st);
void launch_acc_executed(const int* n_edges, double flops_per_edge, double* counter, hipStream_t st);
struct AccBatch { const int* n[32]; double coef[32]; double* dst[32]; int count; };   // (heads.hip)
void launch_acc_batch(const AccBatch& b, hipStream_t st);
// HBM bytes per edge of the FUSED conv (what the kernels of this library have to move): edge record (48 embedding floats, 9 harmonics,
// 3 indices), two gathered 48-float rows for the radial MLP, the gathered D_in-float input row, the D_out-float message
static inline double fused_bytes(int D_in, int D_out) { return 4.0 * (48 + 9 + 3 + 48 + 48 + D_in + D_out); }

// AF2 residue constant tables (data only; generated from the reference's protein_constants.py by tests/golden/make_residue_tables.py)
#define RT_TABLE static const
#include "residue_tables.inc"
#undef RT_TABLE

static thread_local std::string g_err;
void dbfr_set_error(const std::string& s) { g_err = s; }
static int fail(int code, const std::string& s) { g_err = s; return code; }

static thread_local bool g_launch_err = false;
bool dbfr_launch_check(hipError_t e, const char* what) {
  if (e == hipSuccess) return false;
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  g_launch_err = true;
  return true;
}
// entry points call this behind their launches: a launcher that gave up (dbfr_launch_check) makes the call fail with DBFR_ERR_HIP
static int take_launch_error() {
  if (!g_launch_err) return DBFR_OK;
  g_launch_err = false;
  return DBFR_ERR_HIP;
}
int dbfr_current_cu_count() {
  static std::atomic<int> cache[64];                     // (zero-initialised; host threads driving different devices may fill it concurrently)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (!n) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// DBFR_GEMM = f32 | split_f16 | reduce_first: which matrix instruction carries the 144 x W GEMM of the K=144 convs (dbfr_model_set_gemm overrides)
// Anything else -- the retired `split` / `split_l1` / 1 / 2 of rounds 2-4 among it -- is an ERROR (-1; dbfr_model_create fails with DBFR_ERR_ARG): until
// round 6 an unknown value fell silently to the fp32 instruction, a quarter of the default's speed.
static int gemm_from_env() {
  const char* e = getenv("DBFR_GEMM");
  if (!e || !*e) return DBFR_GEMM_DEFAULT;
  if (!strcmp(e, "f32") || !strcmp(e, "0")) return DBFR_GEMM_F32;
  if (!strcmp(e, "split_f16") || !strcmp(e, "3")) return DBFR_GEMM_SPLIT_F16;
  if (!strcmp(e, "reduce_first") || !strcmp(e, "4")) return DBFR_GEMM_REDUCE_FIRST;
  return -1;
}

struct dbfr_model {
  dbfr_model_cfg cfg;
  std::vector<void*> allocs;
  ConvW layer[8][4];   // [l][family]: 0 lig, 1 cross_al, 2 atom, 3 cross_la
  ConvW final_conv, tor_conv, sc_conv;
  ConvW2 layer2[8][4], tor_conv2, sc_conv2;   // k_conv2 layouts of the K=144 convs
  ConvW2 layer2v[8][4];                       // ... of their l = 1 outputs alone (DBFR_GEMM_REDUCE_FIRST: the scalar outputs go through k_convz)
  ConvZ layerz[8][4], tor_convz, sc_convz;    // reduce-first form of the scalar-output paths (convz.hip)
  int use_conv2;
  int gemm_split;      // DBFR_GEMM_*; non-zero: the 144 x W GEMM of the K=144 convs runs on the bf16 matrix pipe with 3-piece operands (conv2s.hip), any batch size
  int conv_fuse;       // big batches: the four convs of a layer as one k_conv grid (conv.hip: k_conv_layer)
  int conv2_layers;    // big batches: interaction layers [0, conv2_layers) still go through k_conv2 (their short W2 favours it)
  int* queue;          // [2] unit queue of k_conv2 (re-armed by the kernel itself)
  std::string fallback_convs;   // ';'-separated names of the convs whose weights two fp16 pieces cannot hold (dbfr_model_fallback_convs)
  std::string rowscaled_convs;  // 'name:depth;' of the convs packed with per-row factors (dbfr_model_rowscaled_convs)
  uint32_t layer_fallback;      // bit l: interaction layer l goes through the fp32-instruction kernel k_conv2 whatever the mode (a bias 2^48 above its row); bit 31: the torsion heads
  int* edge_log; int edge_log_steps, edge_log_graphs;   // dbfr_model_set_edge_log: caller-owned device buffer [steps][6][graphs], or null
  int* tie_log; int tie_log_steps, tie_log_graphs; float tie_tol;   // dbfr_model_set_tie_log: likewise, candidate pairs within tie_tol of a cutoff
  Mlp2 lig_node_emb, lig_edge_emb, atom_edge_emb, la_edge_emb, center_edge_emb, tor_edge_emb, sc_edge_emb;
  Mlp2 tr_final, rot_final, tor_final, sc_final;
  const float* atom_emb[5]; int atom_dims[5];
  const float* atom_lin_t;
  const float *gs_lig_off, *gs_lig_c, *gs_atom_off, *gs_atom_c, *gs_cross_off, *gs_cross_c, *gs_center_off, *gs_center_c;
  int* a14_group;
  // profiling of the dominant kernel
  int profile;
  std::vector<hipEvent_t> ev;
  size_t ev_used;
  double* flops_dev;
  double fused_bytes_last;   // third counter as of the last dbfr_profile_read (before its reset)
  double executed_last;      // fourth counter (flops the matrix pipe executed in the K=144 conv launches), likewise
  double useful_last;        // fifth: the executed flops that are not padding
  double form_bytes_last;    // sixth: HBM bytes the form that runs has to move (DBFR_GEMM_REDUCE_FIRST: scalar-output message columns once per segment, inputs read by both kernels)
  double conv_ms_acc; int64_t conv_launches_acc;
  // side streams for small batches: the four convs of an interaction layer (and the three heads) are independent
  hipStream_t side[3];
  hipEvent_t ev_fork, ev_a, ev_b, ev_join[3];
  bool streams_ready;
};

template <typename T>
static T* upload(dbfr_model* m, const std::vector<T>& h, int* rc) {
  void* d = nullptr;
  size_t bytes = std::max<size_t>(h.size() * sizeof(T), 16);
  if (hipMalloc(&d, bytes) != hipSuccess) { *rc = DBFR_ERR_HIP; g_err = "hipMalloc failed (model weights)"; return nullptr; }
  if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
    *rc = DBFR_ERR_HIP; g_err = "hipMemcpy failed (model weights)"; return nullptr;
  }
  m->allocs.push_back(d);
  return (T*)d;
}

typedef std::map<std::string, const dbfr_tensor*> TMap;

static const float* need(const TMap& tm, const std::string& name, int64_t numel, int* rc) {
  auto it = tm.find(name);
  if (it == tm.end()) { *rc = fail(DBFR_ERR_ARG, "missing tensor '" + name + "'"); return nullptr; }
  if (it->second->numel != numel) {
    *rc = fail(DBFR_ERR_ARG, "tensor '" + name + "' has " + std::to_string(it->second->numel) + " elements, expected " +
                                 std::to_string(numel));
    return nullptr;
  }
  return it->second->data;
}

static int pack_mlp(dbfr_model* m, const TMap& tm, const std::string& name, int in, int hid, int out, bool bias, Mlp2* o) {
  int rc = 0;
  const float* w0 = need(tm, name + ".lin.0.weight", (int64_t)hid * in, &rc);
  const float* w1 = need(tm, name + ".lin.3.weight", (int64_t)out * hid, &rc);
  const float *b0 = nullptr, *b1 = nullptr;
  if (bias) { b0 = need(tm, name + ".lin.0.bias", hid, &rc); b1 = need(tm, name + ".lin.3.bias", out, &rc); }
  if (rc) return rc;
  std::vector<float> w0t((size_t)in * hid), w1t((size_t)hid * out);
  for (int h = 0; h < hid; ++h) for (int i = 0; i < in; ++i) w0t[(size_t)i * hid + h] = w0[(size_t)h * in + i];
  for (int q = 0; q < out; ++q) for (int h = 0; h < hid; ++h) w1t[(size_t)h * out + q] = w1[(size_t)q * hid + h];
  o->in = in; o->hid = hid; o->out = out;
  o->w0t = upload(m, w0t, &rc); o->w1t = upload(m, w1t, &rc);
  o->b0 = o->b1 = nullptr;
  if (bias) {
    o->b0 = upload(m, std::vector<float>(b0, b0 + hid), &rc);
    o->b1 = upload(m, std::vector<float>(b1, b1 + out), &rc);
  }
  return rc;
}

static int pack_conv(dbfr_model* m, const TMap& tm, const std::string& name, int kind, ConvW* o) {
  ConvSpec sp = make_conv_spec(kind);
  for (auto& p : sp.paths)
    if (p.type < 0) return fail(DBFR_ERR_ARG, "unsupported tensor-product path in " + name);
  const int K = sp.K;
  int rc = 0;
  const float* W1 = need(tm, name + ".fc.lin.0.weight", (int64_t)K * K, &rc);
  const float* B1 = need(tm, name + ".fc.lin.0.bias", K, &rc);
  const float* W2 = need(tm, name + ".fc.lin.3.weight", (int64_t)sp.W * K, &rc);
  const float* B2 = need(tm, name + ".fc.lin.3.bias", sp.W, &rc);
  int nirr = 0, n0e = 0;
  for (auto& ir : sp.out) { nirr += ir.mul; if (ir.l == 0 && ir.p == 1) n0e += ir.mul; }
  const float* ms = need(tm, name + ".batch_norm.mean_shift", nirr, &rc);
  const float* aw = need(tm, name + ".batch_norm.affine_weight", nirr, &rc);
  const float* ab = need(tm, name + ".batch_norm.affine_bias", n0e, &rc);
  if (rc) return rc;
  const int KT = K / 16;
  // lin.0 in MFMA A-fragment order: [m][s4][lane][q] = W1[16m + (lane&15)][16 s4 + 4 (lane>>4) + q].  The reduction index is
  // permuted inside every 16-group (MFMA k-step q of lane group g takes k = 16 s4 + 4 g + q) so that the four B operands a lane
  // needs for one A fragment are 4 consecutive floats of its edge's activation row: one ds_read_b128 instead of four ds_read_b32
  std::vector<float> w1p((size_t)KT * KT * 64 * 4);
  for (int mt = 0; mt < KT; ++mt)
    for (int s4 = 0; s4 < KT; ++s4)
      for (int lane = 0; lane < 64; ++lane)
        for (int q = 0; q < 4; ++q)
          w1p[(((size_t)mt * KT + s4) * 64 + lane) * 4 + q] = W1[(size_t)(16 * mt + (lane & 15)) * K + 16 * s4 + 4 * (lane >> 4) + q];
  // lin.3 rows re-ordered for OWNER accumulation (no atomics, reproducible):
  //   an output channel (io, w) is a "pair"; its contributions come from every path into io, all u_in.
  //   4 pairs with the same (type, mul1) path sequence form a group processed in lock step: tile =
  //   4 quads (one per pair = per MFMA lane group) x the same path and u-quad => path type is tile-uniform.
  //   A lane keeps the running message element in registers over the group's tiles and stores it once.
  //   Groups are dealt to the 4 waves of the workgroup by LPT; tiles are laid out wave-major.
  struct Row { int orig; float scale; };
  struct PairPath { const PathDesc* p; };
  struct Pair { int io, w; std::vector<const PathDesc*> paths; };
  std::vector<Pair> pairs;
  for (int io = 0; io < (int)sp.out.size(); ++io)
    for (int w = 0; w < sp.out[io].mul; ++w) {
      Pair pr{io, w, {}};
      for (auto& p : sp.paths) if (p.io == io) pr.paths.push_back(&p);
      std::stable_sort(pr.paths.begin(), pr.paths.end(),
                       [](const PathDesc* a, const PathDesc* b) { return a->type != b->type ? a->type < b->type : a->mul1 > b->mul1; });
      pairs.push_back(pr);
    }
  auto sig = [](const Pair& a) {
    std::string k;
    for (auto* p : a.paths) k += std::to_string(p->type) + ":" + std::to_string(p->mul1) + ",";
    return k;
  };
  struct Group { std::vector<int> pair_idx; int tiles; };
  std::vector<Group> groups;
  {
    std::map<std::string, std::vector<int>> cls;
    std::vector<std::string> order;
    for (int i = 0; i < (int)pairs.size(); ++i) {
      std::string k = sig(pairs[i]);
      if (!cls.count(k)) order.push_back(k);
      cls[k].push_back(i);
    }
    for (auto& k : order) {
      auto& v = cls[k];
      for (size_t i = 0; i < v.size(); i += 4) {
        Group g;
        for (size_t j = i; j < i + 4; ++j) g.pair_idx.push_back(j < v.size() ? v[j] : -1);
        g.tiles = 0;
        for (auto* p : pairs[v[i]].paths) {
          if (p->mul1 % 4) return fail(DBFR_ERR_ARG, "input multiplicity not a multiple of 4 in " + name);
          g.tiles += p->mul1 / 4;
        }
        groups.push_back(g);
      }
    }
  }
  // LPT deal to the 4 waves
  std::vector<int> gorder(groups.size());
  for (size_t i = 0; i < gorder.size(); ++i) gorder[i] = (int)i;
  std::stable_sort(gorder.begin(), gorder.end(), [&](int a, int b) { return groups[a].tiles > groups[b].tiles; });
  std::vector<std::vector<int>> wave_groups(4);
  int wave_load[4] = {0, 0, 0, 0};
  for (int gi : gorder) {
    int best = 0;
    for (int v = 1; v < 4; ++v) if (wave_load[v] < wave_load[best]) best = v;
    wave_groups[best].push_back(gi);
    wave_load[best] += groups[gi].tiles;
  }
  // local search on top of LPT: move / swap groups between the heaviest wave and the others while that lowers the
  // maximum (e.g. W=7776: 24 groups of 15 tiles + 6 of 21 => LPT 126/126/117/117, after refinement 123/123/120/120)
  for (int iter = 0; iter < 64; ++iter) {
    int hi = 0;
    for (int v = 1; v < 4; ++v) if (wave_load[v] > wave_load[hi]) hi = v;
    bool improved = false;
    for (int v = 0; v < 4 && !improved; ++v) {
      if (v == hi) continue;
      for (size_t i = 0; i < wave_groups[hi].size() && !improved; ++i) {
        const int gi = wave_groups[hi][i], ti = groups[gi].tiles;
        // move
        if (std::max(wave_load[hi] - ti, wave_load[v] + ti) < wave_load[hi]) {
          wave_groups[v].push_back(gi); wave_groups[hi].erase(wave_groups[hi].begin() + i);
          wave_load[hi] -= ti; wave_load[v] += ti; improved = true; break;
        }
        for (size_t j = 0; j < wave_groups[v].size(); ++j) {   // swap
          const int gj = wave_groups[v][j], tj = groups[gj].tiles;
          if (tj < ti && std::max(wave_load[hi] - ti + tj, wave_load[v] + ti - tj) < wave_load[hi]) {
            std::swap(wave_groups[hi][i], wave_groups[v][j]);
            wave_load[hi] += tj - ti; wave_load[v] += ti - tj; improved = true; break;
          }
        }
      }
    }
    if (!improved) break;
  }
  std::vector<Row> rows;
  std::vector<RunDesc> runs;
  int wave_tile0[5] = {0, 0, 0, 0, 0}, wave_run0[5] = {0, 0, 0, 0, 0};
  for (int v = 0; v < 4; ++v) {
    std::sort(wave_groups[v].begin(), wave_groups[v].end());
    wave_tile0[v] = (int)rows.size() / 16;
    wave_run0[v] = (int)runs.size();
    for (int gi : wave_groups[v]) {
      const Group& G = groups[gi];
      const Pair& lead = pairs[G.pair_idx[0]];
      for (size_t pi = 0; pi < lead.paths.size(); ++pi) {
        const PathDesc* lp = lead.paths[pi];
        RunDesc rd;
        const uint32_t flags = (pi == 0 ? 1u : 0u) | (pi + 1 == lead.paths.size() ? 2u : 0u);
        const uint32_t tile0 = (uint32_t)rows.size() / 16, nt = (uint32_t)lp->mul1 / 4, x_step = 4u * (2 * lp->l1 + 1);
        if (tile0 >= (1u << 20) || nt >= (1u << 12)) return fail(DBFR_ERR_ARG, "conv too large for the run descriptor");
        rd.tile0_n = tile0 | (nt << 20);
        rd.meta = (uint32_t)lp->type | (flags << 4) | ((uint32_t)lp->sh_off << 8) | (x_step << 12);
        rd.x_off4 = rd.o_off4 = 0;
        for (int g = 0; g < 4; ++g) {
          const int pidx = G.pair_idx[g];
          const PathDesc* p = pidx >= 0 ? pairs[pidx].paths[pi] : lp;
          if (p->type != lp->type || p->sh_off != lp->sh_off || p->mul1 != lp->mul1)
            return fail(DBFR_ERR_ARG, "channel group with non-uniform paths in " + name);
          const int d_o = 2 * p->lo + 1;
          const uint32_t xo = p->in_off;
          const uint32_t oo = pidx >= 0 ? p->out_off + pairs[pidx].w * d_o : (uint32_t)sp.D_out;  // dummy -> trash column
          if (xo > 255 || oo > 255) return fail(DBFR_ERR_ARG, "irreps too wide for the run descriptor");
          rd.x_off4 |= xo << (8 * g);
          rd.o_off4 |= oo << (8 * g);
        }
        runs.push_back(rd);
        for (int u0 = 0; u0 < lp->mul1; u0 += 4)
          for (int g = 0; g < 4; ++g) {
            const int pidx = G.pair_idx[g];
            const PathDesc* p = pidx >= 0 ? pairs[pidx].paths[pi] : lp;
            for (int r = 0; r < 4; ++r)
              rows.push_back(pidx >= 0 ? Row{p->w_off + (u0 + r) * p->mulo + pairs[pidx].w, p->fold} : Row{-1, 0.f});
          }
      }
    }
  }
  wave_run0[4] = (int)runs.size();
  wave_tile0[4] = (int)rows.size() / 16;
  const int n_tiles = (int)rows.size() / 16;
  std::vector<float> w2p((size_t)n_tiles * KT * 64 * 4), b2p((size_t)n_tiles * 16);
  for (int t = 0; t < n_tiles; ++t) {
    for (int i = 0; i < 16; ++i) { const Row& r = rows[16 * t + i]; b2p[16 * t + i] = r.orig >= 0 ? r.scale * B2[r.orig] : 0.f; }
    for (int s4 = 0; s4 < KT; ++s4)
      for (int lane = 0; lane < 64; ++lane) {
        const Row& r = rows[16 * t + (lane & 15)];
        for (int q = 0; q < 4; ++q)
          w2p[(((size_t)t * KT + s4) * 64 + lane) * 4 + q] =
              r.orig >= 0 ? r.scale * W2[(size_t)r.orig * K + 16 * s4 + 4 * (lane >> 4) + q] : 0.f;
      }
  }
  o->K = K; o->D_in = sp.D_in; o->D_out = sp.D_out; o->n_tiles = n_tiles; o->W = sp.W;
  for (int v = 0; v < 5; ++v) { o->wave_tile0[v] = wave_tile0[v]; o->wave_run0[v] = wave_run0[v]; }
  o->W1p = upload(m, w1p, &rc);
  o->b1 = upload(m, std::vector<float>(B1, B1 + K), &rc);
  o->W2p = upload(m, w2p, &rc);
  o->b2p = upload(m, b2p, &rc);
  o->runs = upload(m, runs, &rc);
  LNDesc& ln = o->ln;
  memset(&ln, 0, sizeof ln);
  ln.nblk = (int)sp.out.size();
  int off = 0;
  for (int i = 0; i < ln.nblk; ++i) {
    ln.mul[i] = sp.out[i].mul; ln.dim[i] = sp.out[i].dim(); ln.off[i] = off;
    ln.is0e[i] = sp.out[i].l == 0 && sp.out[i].p == 1;
    off += ln.mul[i] * ln.dim[i];
  }
  ln.mean_shift = upload(m, std::vector<float>(ms, ms + nirr), &rc);
  ln.weight = upload(m, std::vector<float>(aw, aw + nirr), &rc);
  ln.bias = upload(m, std::vector<float>(ab, ab + std::max(n0e, 0)), &rc);
  return rc;
}

// Tiles [tile0, tile0 + nt) of a [tile][K/16][64][4] fp32 fragment array (K = 144) -> the k_conv2h tile format, all with ONE
// power-of-two factor 2^k (returned).  hi = fp16(v 2^k), lo = fp16(v 2^k - hi), both rounded to nearest even: 2 x 11 significand
// bits + the sign of lo = 23 of fp32's 24.  fp16 has five exponent bits, so 2^k puts the largest |v| of these tiles into
// [2^14, 2^15): pieces of every value within 2^-17 of that maximum stay exact to 22 bits (fp16 subnormals reach down to 2^-24 and
// the matrix pipe keeps them -- tools/exp/split_f16.hip part D), smaller ones keep an absolute error of 2^-40 of the maximum.
// Per tile: [2 pieces][4 k-steps of 32][64][8] -- step s = fp32 k-steps 2s and 2s+1 of the same lane -- then [64 lanes][hi 4 | lo 4]
// for the last 16 k (one 16-byte fragment per lane: the A operand of the x32 MFMA that carries both small products of those 16 k,
// its first half the A operand of the x16 MFMA of the large one), then the tile's 16 bias values (fp32) x 2^k = 9280 B.
#define CH_TILE_BYTES_HOST 9280
// *depth (may be null) receives the largest d over the rows of these tiles with row maximum 2^(15 - d) after scaling (0 for the row that
// sets the factor): a row with d <= F16_ROW_DEPTH_OK keeps 22 significant bits in its two pieces, a deeper one loses one bit per step
// (relative error 2^(d - 40) of the row's own dot product) -- the model routes such a conv to the three-bf16-piece kernel (bf16 has
// fp32's exponent range), see fallback_convs.
#define F16_ROW_DEPTH_OK 17
// rinv (may be null): per-ROW factors on top of 2^k -- every row of these tiles is multiplied by the power of two 2^d that puts ITS largest
// magnitude into [2^14, 2^15) (as far as its bias allows), rinv[16 t + row] = 2^-d is what the kernel takes off the accumulator row
// (k_conv2h<.., ROWF>; the hidden layers of k_conv2h / k_convz for W1h).  No row is deep any more: *depth reports what is left (bias-bound rows).
static int pack_f16_tiles(const float* frag, const float* bias16, int tile0, int nt, uint16_t* out, int* depth = nullptr, float* rinv = nullptr) {
  constexpr int KT = 9;
  const size_t tile_h = CH_TILE_BYTES_HOST / 2, tail_off = 8192 / 2, bias_off = 9216 / 2;
  auto h16 = [](float v) { const _Float16 h = (_Float16)v; uint16_t u; memcpy(&u, &h, 2); return u; };
  float mx = 0.f;
  for (size_t i = (size_t)tile0 * KT * 256; i < (size_t)(tile0 + nt) * KT * 256; ++i) mx = std::max(mx, fabsf(frag[i]));
  int e = 0;
  if (mx > 0.f) (void)frexpf(mx, &e);            // mx = f 2^e, f in [0.5, 1)
  int k = mx > 0.f ? std::max(-120, std::min(120, 15 - e)) : 0;
  // the bias rows carry 2^k too and meet the per-edge factor (at most 2^64, conv2h.hip) in the accumulator, and the accumulator meets
  // the input features in the contraction: keep |bias| 2^k below 2^48 so that all of it stays a finite fp32 number however small the
  // weights of the run are next to its bias (never binding for weights and biases of comparable size; a run it binds for loses
  // relative precision in its pieces, nothing else)
  float bmx = 0.f;
  for (int i = 16 * tile0; i < 16 * (tile0 + nt); ++i) bmx = std::max(bmx, fabsf(bias16[i]));
  if (bmx > 0.f) { int eb = 0; (void)frexpf(bmx, &eb); k = std::min(k, 48 - eb); }
  const float sc = ldexpf(1.f, k);
  std::vector<int> rowd((size_t)nt * 16, 0);        // extra exponent of every row (0 without per-row factors)
  int dmax = 0;
  for (int t = tile0; t < tile0 + nt; ++t)
    for (int row = 0; row < 16; ++row) {
      float rm = 0.f;
      for (int s4 = 0; s4 < KT; ++s4)
        for (int gq = 0; gq < 4; ++gq)
          for (int q = 0; q < 4; ++q) rm = std::max(rm, fabsf(frag[(((size_t)t * KT + s4) * 64 + 16 * gq + row) * 4 + q]));
      if (rm <= 0.f) continue;                     // (all-zero rows: padded channels)
      int er = 0;
      (void)frexpf(rm, &er);
      int d = 15 - (er + k);
      if (rinv) {
        int up = std::max(d, 0);
        const float br = fabsf(bias16[16 * t + row]);
        if (br > 0.f) { int eb = 0; (void)frexpf(br, &eb); up = std::max(0, std::min(up, 48 - eb - k)); }   // (|bias| 2^(k + d) stays below 2^48, as for k)
        up = std::min(up, 100);
        rowd[(size_t)(t - tile0) * 16 + row] = up;
        d -= up;
      }
      dmax = std::max(dmax, d);
    }
  if (depth) *depth = dmax;
  if (rinv)
    for (int i = 0; i < nt * 16; ++i) rinv[(size_t)tile0 * 16 + i] = ldexpf(1.f, -rowd[i]);
  for (int t = tile0; t < tile0 + nt; ++t) {
    float b[16];
    for (int i = 0; i < 16; ++i) b[i] = bias16[16 * t + i] * ldexpf(sc, rowd[(size_t)(t - tile0) * 16 + i]);
    memcpy(&out[t * tile_h + bias_off], b, 64);
    for (int s4 = 0; s4 < KT; ++s4)
      for (int lane = 0; lane < 64; ++lane)
        for (int q = 0; q < 4; ++q) {
          const float v = frag[(((size_t)t * KT + s4) * 64 + lane) * 4 + q] * ldexpf(sc, rowd[(size_t)(t - tile0) * 16 + (lane & 15)]);
          const _Float16 hi = (_Float16)v;
          const uint16_t pc[2] = {h16(v), h16(v - (float)hi)};
          for (int i = 0; i < 2; ++i) {
            if (s4 < 8) out[t * tile_h + ((size_t)(i * 4 + (s4 >> 1)) * 64 + lane) * 8 + 4 * (s4 & 1) + q] = pc[i];
            else out[t * tile_h + tail_off + (size_t)lane * 8 + 4 * i + q] = pc[i];
          }
        }
  }
  return k;
}

// k_conv2 layout (conv2.hip): same channel-owner row order as pack_conv, but ONE tile sequence for all waves.
// vector_only: the rows that feed an l = 0 output irrep are left out (DBFR_GEMM_REDUCE_FIRST serves them through k_convz); only the fp16 pieces
// are packed then, and a conv without l = 1 outputs (torsion convs) gets n_tiles = 0.
static int pack_conv2(dbfr_model* m, const TMap& tm, const std::string& name, int kind, const ConvW& base, ConvW2* o, bool vector_only = false) {
  ConvSpec sp = make_conv_spec(kind);
  const int K = sp.K;
  if (K != 144) return fail(DBFR_ERR_ARG, "k_conv2 layout is for the K=144 convs");
  int rc = 0;
  const float* W2 = need(tm, name + ".fc.lin.3.weight", (int64_t)sp.W * K, &rc);
  const float* B2 = need(tm, name + ".fc.lin.3.bias", sp.W, &rc);
  if (rc) return rc;
  const int KT = K / 16;
  struct Pair { int io, w; std::vector<const PathDesc*> paths; };
  std::vector<Pair> pairs;
  for (int io = 0; io < (int)sp.out.size(); ++io)
    for (int w = 0; w < sp.out[io].mul; ++w) {
      if (vector_only && sp.out[io].l == 0) continue;
      Pair pr{io, w, {}};
      for (auto& p : sp.paths) if (p.io == io) pr.paths.push_back(&p);
      std::stable_sort(pr.paths.begin(), pr.paths.end(),
                       [](const PathDesc* a, const PathDesc* b) { return a->type != b->type ? a->type < b->type : a->mul1 > b->mul1; });
      pairs.push_back(pr);
    }
  auto sig = [](const Pair& a) {
    std::string k;
    for (auto* p : a.paths) k += std::to_string(p->type) + ":" + std::to_string(p->mul1) + ":" + std::to_string(p->in_off) + ",";
    return k;
  };
  struct Group { std::vector<int> pair_idx; int tiles; int xph; };
  std::vector<Group> groups;
  {
    std::map<std::string, std::vector<int>> cls;
    std::vector<std::string> order;
    for (int i = 0; i < (int)pairs.size(); ++i) {
      std::string k = sig(pairs[i]);
      if (!cls.count(k)) order.push_back(k);
      cls[k].push_back(i);
    }
    for (auto& k : order) {
      auto& v = cls[k];
      for (size_t i = 0; i < v.size(); i += 4) {
        Group g;
        for (size_t j = i; j < i + 4; ++j) g.pair_idx.push_back(j < v.size() ? v[j] : -1);
        g.tiles = 0;
        bool hi = false, lo = false;
        for (auto* p : pairs[v[i]].paths) {
          if (p->mul1 % 4) return fail(DBFR_ERR_ARG, "input multiplicity not a multiple of 4 in " + name);
          g.tiles += p->mul1 / 4;
          if (p->in_off >= 120) hi = true;
          if (p->in_off < 48) lo = true;
        }
        // the LDS x row holds 120 floats: [0e|1o|1e] for the groups that read the 48x0e inputs, [0o|1o|1e] for those
        // that read the 48x0o inputs (x[120:168]); no output irrep of this model family reads both
        if (hi && lo) return fail(DBFR_ERR_ARG, "channel group reads both scalar input blocks in " + name);
        g.xph = hi ? 1 : 0;
        groups.push_back(g);
      }
    }
  }
  std::stable_sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) { return a.xph < b.xph; });
  std::vector<int> rows_orig; std::vector<float> rows_scale;
  std::vector<RunDesc> runs;
  std::vector<int> group_run0, group_tiles;
  for (const Group& G : groups) {
    group_run0.push_back((int)runs.size());
    group_tiles.push_back(G.tiles);
    const Pair& lead = pairs[G.pair_idx[0]];
    for (size_t pi = 0; pi < lead.paths.size(); ++pi) {
      const PathDesc* lp = lead.paths[pi];
      RunDesc rd;
      const uint32_t flags = (pi == 0 ? 1u : 0u) | (pi + 1 == lead.paths.size() ? 2u : 0u);
      const uint32_t tile0 = (uint32_t)rows_orig.size() / 16, nt = (uint32_t)lp->mul1 / 4, x_step = 4u * (2 * lp->l1 + 1);
      if (tile0 >= (1u << 20) || nt >= (1u << 12)) return fail(DBFR_ERR_ARG, "conv too large for the run descriptor");
      rd.tile0_n = tile0 | (nt << 20);
      rd.meta = (uint32_t)lp->type | (flags << 4) | ((uint32_t)lp->sh_off << 8) | (x_step << 12) | ((uint32_t)G.xph << 20);
      rd.x_off4 = rd.o_off4 = 0;
      for (int g = 0; g < 4; ++g) {
        const int pidx = G.pair_idx[g];
        const PathDesc* p = pidx >= 0 ? pairs[pidx].paths[pi] : lp;
        if (p->type != lp->type || p->sh_off != lp->sh_off || p->mul1 != lp->mul1 || p->in_off != lp->in_off)
          return fail(DBFR_ERR_ARG, "channel group with non-uniform paths in " + name);
        const int d_o = 2 * p->lo + 1;
        const uint32_t xo = p->in_off >= 120 ? p->in_off - 120 : p->in_off;     // position inside the 120-float LDS row
        const uint32_t oo = pidx >= 0 ? p->out_off + pairs[pidx].w * d_o : 255u;  // 255 >= D_out: padded channel, never stored
        if (xo + 4 * (2 * p->l1 + 1) * (p->mul1 / 4) > 120 || (oo != 255u && (int)oo >= sp.D_out))
          return fail(DBFR_ERR_ARG, "irreps do not fit the k_conv2 row layout in " + name);
        rd.x_off4 |= xo << (8 * g);
        rd.o_off4 |= oo << (8 * g);
      }
      runs.push_back(rd);
      for (int u0 = 0; u0 < lp->mul1; u0 += 4)
        for (int g = 0; g < 4; ++g) {
          const int pidx = G.pair_idx[g];
          const PathDesc* p = pidx >= 0 ? pairs[pidx].paths[pi] : lp;
          for (int r = 0; r < 4; ++r) {
            rows_orig.push_back(pidx >= 0 ? p->w_off + (u0 + r) * p->mulo + pairs[pidx].w : -1);
            rows_scale.push_back(pidx >= 0 ? p->fold : 0.f);
          }
        }
    }
  }
  const int n_tiles = (int)rows_orig.size() / 16;
  std::vector<float> w2q((size_t)n_tiles * KT * 64 * 4), b2q((size_t)n_tiles * 16);
  for (int t = 0; t < n_tiles; ++t) {
    for (int i = 0; i < 16; ++i) { const int ro = rows_orig[16 * t + i]; b2q[16 * t + i] = ro >= 0 ? rows_scale[16 * t + i] * B2[ro] : 0.f; }
    for (int s4 = 0; s4 < KT; ++s4)
      for (int lane = 0; lane < 64; ++lane) {
        const int ro = rows_orig[16 * t + (lane & 15)];
        const float sc = rows_scale[16 * t + (lane & 15)];
        for (int q = 0; q < 4; ++q)
          w2q[(((size_t)t * KT + s4) * 64 + lane) * 4 + q] = ro >= 0 ? sc * W2[(size_t)ro * K + 16 * s4 + 4 * (lane >> 4) + q] : 0.f;
      }
  }
  memset(o, 0, sizeof *o);
  o->D_in = sp.D_in; o->D_out = sp.D_out; o->n_tiles = n_tiles; o->n_runs = (int)runs.size();
  o->W1p = base.W1p; o->b1 = base.b1;
  if (!vector_only) {
  o->W2q = upload(m, w2q, &rc);
  o->b2q = upload(m, b2q, &rc);
  }
  {   // the same fragments cut into TWO fp16 pieces (conv2h.hip), run by run (a run = the tiles of one tensor-product path of one channel group):
      // pack_f16_tiles below.  k travels in RunDesc.meta bits 24..31; the kernel folds 2^-k into the run's harmonics.
    // A conv with a run whose rows lie further apart than two fp16 pieces hold behind ONE factor (depth > F16_ROW_DEPTH_OK) is packed again with
    // per-ROW factors (W2rinv) and served by k_conv2h<.., ROWF>, which takes them off the accumulator rows: four more vector instructions per edge
    // block and tile instead of the six-product kernel (round 4's fall-back).  DBFR_F16_ROWSCALE=0 (developer) keeps the old routing, =2 packs
    // every conv this way.
    const int rowscale = getenv("DBFR_F16_ROWSCALE") ? atoi(getenv("DBFR_F16_ROWSCALE")) : 1;   // (read at every model creation: a test packs one model this way)
    std::vector<uint16_t> w2h((size_t)n_tiles * (CH_TILE_BYTES_HOST / 2) + 512, 0);
    std::vector<float> w2rinv((size_t)std::max(n_tiles, 1) * 16, 1.f);
    int depth_max = 0, depth_run = 0;
    for (int pass = 0; pass < 2; ++pass) {
      const bool rowf = pass == 1;
      if (rowf && !(rowscale == 2 || (rowscale == 1 && depth_max > F16_ROW_DEPTH_OK))) break;
      if (rowf) depth_run = depth_max;
      depth_max = 0;
      for (RunDesc& rd : runs) {
        const int tile0 = rd.tile0_n & 0xfffff, nt = rd.tile0_n >> 20;
        int depth = 0;
        const int k = pack_f16_tiles(w2q.data(), b2q.data(), tile0, nt, w2h.data(), &depth, rowf ? w2rinv.data() : nullptr);
        depth_max = std::max(depth_max, depth);
        rd.meta = (rd.meta & 0x00ffffffu) | ((uint32_t)(k & 0xff) << 24);
      }
      if (!rowf) depth_run = depth_max;
      else o->W2rinv = upload(m, w2rinv, &rc);
    }
    o->W2h = upload(m, w2h, &rc);
    // the hidden layer W1 (144 x 144) the same way: nine 16-row tiles with ONE factor 2^k1
    const float* W1 = need(tm, name + ".fc.lin.0.weight", (int64_t)K * K, &rc);
    const float* B1 = need(tm, name + ".fc.lin.0.bias", K, &rc);
    if (rc) return rc;
    std::vector<float> w1q((size_t)KT * KT * 256);
    for (int mt = 0; mt < KT; ++mt)
      for (int s4 = 0; s4 < KT; ++s4)
        for (int lane = 0; lane < 64; ++lane)
          for (int q = 0; q < 4; ++q)
            w1q[(((size_t)mt * KT + s4) * 64 + lane) * 4 + q] = W1[(size_t)(16 * mt + (lane & 15)) * K + 16 * s4 + 4 * (lane >> 4) + q];
    std::vector<uint16_t> w1h((size_t)KT * (CH_TILE_BYTES_HOST / 2) + 512, 0);
    int depth1 = 0;
    o->k1 = pack_f16_tiles(w1q.data(), B1, 0, KT, w1h.data(), &depth1);
    depth_run = std::max(depth_run, depth1);
    if (rowscale == 2 || (rowscale == 1 && depth1 > F16_ROW_DEPTH_OK)) {
      std::vector<float> w1rinv((size_t)KT * 16, 1.f);
      o->k1 = pack_f16_tiles(w1q.data(), B1, 0, KT, w1h.data(), &depth1, w1rinv.data());
      o->W1rinv = upload(m, w1rinv, &rc);
    }
    o->W1h = upload(m, w1h, &rc);
    o->f16_depth = std::max(depth_max, depth1);
    o->f16_depth_run = depth_run;
  }
  o->runs = upload(m, runs, &rc);
  // contiguous group ranges of near-equal tile count for S = 1, 2, 4, 8
  const int n_g = (int)groups.size();
  int total_tiles = 0;
  for (int t : group_tiles) total_tiles += t;
  for (int si = 0; si < 4; ++si) {
    const int S = 1 << si;
    int gi = 0, acc = 0;
    for (int p = 0; p < S; ++p) {
      o->part_run[si][p] = gi < n_g ? group_run0[gi] : (int)runs.size();
      const long target = (long)total_tiles * (p + 1) / S;
      while (gi < n_g && (acc + group_tiles[gi] / 2 < target || p == S - 1)) { acc += group_tiles[gi]; ++gi; }
    }
    for (int p = S; p <= 8; ++p) o->part_run[si][p] = (int)runs.size();
  }
  return rc;
}

// Reduce-first form of the scalar-output paths (convz.hip; layout: common.h ConvZ).  c tiles of 16 (path, u_in) pairs per output irrep,
// scalar-input paths first (48 = 3 full tiles), then the vector-input ones, padded to a tile; W2' = lin.3 row x fold, bias as k = 144,
// every output ROW (irrep, channel) x its own power of two 2^s (largest |value| into [2^14, 2^15)), cut into two fp16 pieces and stored as the
// A fragments of step B: [c tile][k tile][k-step v][w tile][piece][lane (n' = channel, g')][i] = W2'[c = 16 ct + 4 g' + i % 4,
// k = 16 kt + v + 8 (i / 4), channel 16 wt + n'] -- the order in which step A leaves Z in LDS.
static int pack_convz(dbfr_model* m, const TMap& tm, const std::string& name, int kind, const ConvW2& base, ConvZ* o) {
  ConvSpec sp = make_conv_spec(kind);
  const int K = sp.K;
  memset(o, 0, sizeof *o);
  if (K != 144) return fail(DBFR_ERR_ARG, "k_convz is for the K=144 convs");
  int rc = 0;
  const float* W2 = need(tm, name + ".fc.lin.3.weight", (int64_t)sp.W * K, &rc);
  const float* B2 = need(tm, name + ".fc.lin.3.bias", sp.W, &rc);
  if (rc) return rc;
  struct Col { const PathDesc* p; int u; };
  std::vector<std::vector<Col>> cols;      // per scalar output irrep, padded to tiles (p == nullptr: padding)
  std::vector<int> ios;
  for (int io = 0; io < (int)sp.out.size(); ++io) {
    if (sp.out[io].l != 0) continue;
    if (sp.out[io].mul != NS) return fail(DBFR_ERR_ARG, "scalar output irrep with a multiplicity other than 48 in " + name);
    std::vector<Col> c;
    for (int pass = 0; pass < 2; ++pass)
      for (auto& p : sp.paths) {
        if (p.io != io || p.l1 != pass) continue;
        if (p.l1 != p.l2 || p.lo != 0 || p.l1 > 1) return fail(DBFR_ERR_ARG, "unsupported scalar-output path in " + name);
        for (int u = 0; u < p.mul1; ++u) c.push_back({&p, u});
      }
    int n_scalar = 0;
    for (auto& q : c) n_scalar += q.p->l1 == 0;
    if (n_scalar % 16) return fail(DBFR_ERR_ARG, "scalar-input columns do not fill whole tiles in " + name);
    while (c.size() % 16) c.push_back({nullptr, 0});
    cols.push_back(c);
    ios.push_back(io);
  }
  if (ios.size() > 2) return fail(DBFR_ERR_ARG, "more than two scalar output irreps in " + name);
  o->n_io = (int)ios.size();
  int nct_total = 0;
  for (int i = 0; i < o->n_io; ++i) {
    o->nct[i] = (int)cols[i].size() / 16;
    for (auto& q : cols[i]) o->nc_valid[i] += q.p != nullptr;
    o->ct0[i] = nct_total;
    nct_total += o->nct[i];
    int off = 0;
    for (int j = 0; j < ios[i]; ++j) off += sp.out[j].mul * sp.out[j].dim();
    o->out_off[i] = off;
  }
  if (nct_total > CZ_MAXCT) return fail(DBFR_ERR_ARG, "too many c tiles in " + name);
  std::vector<uint32_t> cdesc((size_t)std::max(nct_total, 1) * 16, 0u);
  std::vector<float> rowinv((size_t)std::max(o->n_io, 1) * NS, 1.f);
  std::vector<uint16_t> w2z((size_t)nct_total * CZ_NKT * 8 * (CZ_TILE_BYTES / 2) + 512, 0);
  auto h16 = [](float v) { const _Float16 h = (_Float16)v; uint16_t u; memcpy(&u, &h, 2); return u; };
  for (int i = 0; i < o->n_io; ++i) {
    const auto& c = cols[i];
    for (size_t j = 0; j < c.size(); ++j)
      if (c[j].p) cdesc[(size_t)o->ct0[i] * 16 + j] = (uint32_t)(c[j].p->in_off + c[j].u * (2 * c[j].p->l1 + 1)) | ((uint32_t)c[j].p->l1 << 12) |
                                                      ((uint32_t)c[j].p->sh_off << 16) | 0x80000000u;
    auto wval = [&](int cc, int k, int w) -> float {    // W2'[c, k, w]
      if (cc >= (int)c.size() || !c[cc].p || k > K) return 0.f;
      const PathDesc* p = c[cc].p;
      const size_t row = (size_t)p->w_off + (size_t)c[cc].u * p->mulo + w;
      return p->fold * (k < K ? W2[row * K + k] : B2[row]);
    };
    for (int w = 0; w < NS; ++w) {
      float mx = 0.f;
      for (int cc = 0; cc < (int)c.size(); ++cc)
        for (int k = 0; k <= K; ++k) mx = std::max(mx, fabsf(wval(cc, k, w)));
      int s = 0;
      if (mx > 0.f) { int e = 0; (void)frexpf(mx, &e); s = std::max(-100, std::min(100, 15 - e)); }
      rowinv[(size_t)i * NS + w] = ldexpf(1.f, -s);
      const float sc = ldexpf(1.f, s);
      const int wt = w >> 4, np = w & 15;
      for (int ct = 0; ct < o->nct[i]; ++ct)
        for (int kt = 0; kt < CZ_NKT; ++kt)
          for (int v = 0; v < 8; ++v)
            for (int gp = 0; gp < 4; ++gp)
              for (int t = 0; t < 8; ++t) {
                const float val = wval(16 * ct + 4 * gp + (t & 3), 16 * kt + v + 8 * (t >> 2), w) * sc;   // Z slot (v, g'' = gp, t): k_local = v + 8 (t / 4), c_local = 4 g'' + t % 4 (convz.hip)
                const _Float16 hi = (_Float16)val;
                const size_t tile = ((size_t)(o->ct0[i] + ct) * CZ_NKT + kt) * 8 + v;
                const size_t base_h = tile * (CZ_TILE_BYTES / 2) + (size_t)(wt * 2) * 512 + (size_t)(16 * gp + np) * 8 + t;
                w2z[base_h] = h16(val);
                w2z[base_h + 512] = h16(val - (float)hi);
              }
    }
  }
  o->cdesc = upload(m, cdesc, &rc);
  o->rowinv = upload(m, rowinv, &rc);
  o->W2z = upload(m, w2z, &rc);
  o->W1h = base.W1h; o->k1 = base.k1; o->W1rinv = base.W1rinv;
  return rc;
}

extern "C" int dbfr_test_pack_f16_tiles(const float* frag, const float* bias, int32_t n_tiles, void* out, int32_t* k_out) {
  if (!frag || !bias || !out || !k_out || n_tiles <= 0) return fail(DBFR_ERR_ARG, "dbfr_test_pack_f16_tiles: bad argument");
  *k_out = pack_f16_tiles(frag, bias, 0, n_tiles, (uint16_t*)out);
  return DBFR_OK;
}

extern "C" int dbfr_test_pack_f16_rows(const float* frag, const float* bias, int32_t n_tiles, void* out, int32_t* k_out, float* rinv_out, int32_t* depth_out) {
  if (!frag || !bias || !out || !k_out || !rinv_out || n_tiles <= 0) return fail(DBFR_ERR_ARG, "dbfr_test_pack_f16_rows: bad argument");
  int depth = 0;
  *k_out = pack_f16_tiles(frag, bias, 0, n_tiles, (uint16_t*)out, &depth, rinv_out);
  if (depth_out) *depth_out = depth;
  return DBFR_OK;
}

extern "C" int dbfr_test_pack_f16_depth(const float* frag, const float* bias, int32_t n_tiles, int32_t* depth_out, int32_t* depth_ok_out) {
  if (!frag || !bias || !depth_out || n_tiles <= 0) return fail(DBFR_ERR_ARG, "dbfr_test_pack_f16_depth: bad argument");
  std::vector<uint16_t> out((size_t)n_tiles * (CH_TILE_BYTES_HOST / 2) + 512, 0);
  int depth = 0;
  (void)pack_f16_tiles(frag, bias, 0, n_tiles, out.data(), &depth);
  *depth_out = depth;
  if (depth_ok_out) *depth_ok_out = F16_ROW_DEPTH_OK;
  return DBFR_OK;
}

extern "C" int dbfr_model_rowscaled_convs(const dbfr_model* m, char* names, size_t names_cap) {
  if (!m) return fail(DBFR_ERR_ARG, "null model");
  int n = 0;
  for (char c : m->rowscaled_convs) n += c == ';';
  if (names && names_cap) {
    const size_t len = std::min(names_cap - 1, m->rowscaled_convs.size());
    memcpy(names, m->rowscaled_convs.data(), len);
    names[len] = 0;
  }
  return n;
}

extern "C" int dbfr_model_fallback_convs(const dbfr_model* m, char* names, size_t names_cap) {
  if (!m) return fail(DBFR_ERR_ARG, "null model");
  int n = 0;
  for (char c : m->fallback_convs) n += c == ';';
  if (names && names_cap) {
    const size_t len = std::min(names_cap - 1, m->fallback_convs.size());
    memcpy(names, m->fallback_convs.data(), len);
    names[len] = 0;
  }
  return n;
}

// the per-graph read-outs are laid out for a graph capacity: a batch with MORE graphs is refused here, before any launch (a smaller one -- the ragged last
// batch of a sharded run -- fills the first G entries of each row)
static int check_logs(const dbfr_model* m, int G) {
  if (m->edge_log && G > m->edge_log_graphs) return fail(DBFR_ERR_ARG, "dbfr_model_set_edge_log: the buffer was sized for " + std::to_string(m->edge_log_graphs) + " graphs, this batch has " + std::to_string(G));
  if (m->tie_log && G > m->tie_log_graphs) return fail(DBFR_ERR_ARG, "dbfr_model_set_tie_log: the buffer was sized for " + std::to_string(m->tie_log_graphs) + " graphs, this batch has " + std::to_string(G));
  return DBFR_OK;
}

extern "C" int dbfr_model_set_tie_log(dbfr_model* m, int32_t* log_dev, int32_t n_steps_cap, int32_t n_graphs_cap, float tol) {
  if (!m || (log_dev && (n_steps_cap <= 0 || n_graphs_cap <= 0 || !(tol > 0.f)))) return fail(DBFR_ERR_ARG, "dbfr_model_set_tie_log: bad argument");
  m->tie_log = log_dev; m->tie_log_steps = log_dev ? n_steps_cap : 0; m->tie_log_graphs = log_dev ? n_graphs_cap : 0; m->tie_tol = log_dev ? tol : 0.f;
  return DBFR_OK;
}

extern "C" int dbfr_model_set_edge_log(dbfr_model* m, int32_t* log_dev, int32_t n_steps_cap, int32_t n_graphs_cap) {
  if (!m || (log_dev && (n_steps_cap <= 0 || n_graphs_cap <= 0))) return fail(DBFR_ERR_ARG, "dbfr_model_set_edge_log: bad argument");
  m->edge_log = log_dev; m->edge_log_steps = log_dev ? n_steps_cap : 0; m->edge_log_graphs = log_dev ? n_graphs_cap : 0;
  return DBFR_OK;
}

extern "C" int dbfr_model_set_gemm(dbfr_model* m, int32_t mode) {
  if (!m || mode < DBFR_GEMM_F32 || mode > DBFR_GEMM_REDUCE_FIRST || mode == 1 || mode == 2) return fail(DBFR_ERR_ARG, "dbfr_model_set_gemm: bad argument (modes 1 and 2, the three-bf16-piece kernels, were retired with ABI 5 / 4)");
  m->gemm_split = mode;
  return DBFR_OK;
}
extern "C" int dbfr_model_get_gemm(const dbfr_model* m) { return m ? m->gemm_split : DBFR_ERR_ARG; }

extern "C" int dbfr_abi_version(void) { return DBFR_ABI_VERSION; }
#ifndef DBFR_BUILD_ID
#define DBFR_BUILD_ID "unknown"
#endif
extern "C" const char* dbfr_build_id(void) { return DBFR_BUILD_ID; }
extern "C" const char* dbfr_last_error(void) { return g_err.c_str(); }

extern "C" int dbfr_wigner3j(int32_t l1, int32_t l2, int32_t l3, double* out) {
  if (!out || l1 < 0 || l2 < 0 || l3 < 0 || l1 > 4 || l2 > 4 || l3 > 4) return fail(DBFR_ERR_ARG, "bad l");
  std::vector<double> v;
  wigner3j_real(l1, l2, l3, v);
  memcpy(out, v.data(), v.size() * sizeof(double));
  return DBFR_OK;
}

extern "C" int dbfr_conv_paths(int32_t kind, int32_t* t10, int32_t max_paths, int32_t* weight_numel) {
  if (kind < 0 || kind > 5) return fail(DBFR_ERR_ARG, "bad conv kind");
  ConvSpec sp = make_conv_spec(kind);
  if (weight_numel) *weight_numel = sp.W;
  int n = (int)sp.paths.size();
  for (int i = 0; i < n && i < max_paths && t10; ++i) {
    const PathDesc& p = sp.paths[i];
    int32_t* r = t10 + 10 * i;
    r[0] = p.i1; r[1] = p.i2; r[2] = p.io; r[3] = p.l1; r[4] = p.l2; r[5] = p.lo; r[6] = p.mul1; r[7] = p.mulo; r[8] = p.w_off;
    memcpy(&r[9], &p.coeff, 4);
  }
  return n;
}

static int model_create_impl(const dbfr_model_cfg* cfg, const dbfr_tensor* tensors, int32_t n_tensors, dbfr_model** out);
// DBFR_F16_DEPTH (developer): row depth above which a conv leaves the fp16 kernel; 1000 = never (measures what the fall-back protects from)
static int f16_depth_ok() { static const int v = getenv("DBFR_F16_DEPTH") ? atoi(getenv("DBFR_F16_DEPTH")) : F16_ROW_DEPTH_OK; return v; }

extern "C" int dbfr_model_create(const dbfr_model_cfg* cfg, const dbfr_tensor* tensors, int32_t n_tensors,
                                 dbfr_model** out) {
  try {   // weight re-tiling allocates on the host: no exception crosses the C ABI
    return model_create_impl(cfg, tensors, n_tensors, out);
  } catch (const std::exception& e) {
    return fail(DBFR_ERR_ARG, std::string("dbfr_model_create: ") + e.what());
  }
}

static int model_create_impl(const dbfr_model_cfg* cfg, const dbfr_tensor* tensors, int32_t n_tensors, dbfr_model** out) {
  if (!cfg || !tensors || !out) return fail(DBFR_ERR_ARG, "null argument");
  if (cfg->ns != NS || cfg->nv != NV || cfg->sh_lmax != 2 || cfg->distance_embed_dim != EMB ||
      cfg->sigma_embed_dim != EMB || cfg->num_conv_layers < 1 || cfg->num_conv_layers > 8)
    return fail(DBFR_ERR_ARG, "unsupported model configuration (need ns=48, nv=12, sh_lmax=2, 32-d embeddings)");
  if (cfg->lig_edge_features + 2 * EMB > 80 || cfg->lig_node_features + EMB > 80)
    return fail(DBFR_ERR_ARG, "ligand feature width too large");
  std::string serr;
  int rc = so3_selftest(serr);
  if (rc) return fail(rc, serr);
  TMap tm;
  for (int i = 0; i < n_tensors; ++i) tm[tensors[i].name] = &tensors[i];
  dbfr_model* m = new dbfr_model();
  m->cfg = *cfg;
  m->profile = 0; m->ev_used = 0; m->flops_dev = nullptr; m->fused_bytes_last = 0; m->executed_last = 0; m->useful_last = 0; m->form_bytes_last = 0; m->conv_ms_acc = 0; m->conv_launches_acc = 0;
  m->streams_ready = false;
  m->edge_log = nullptr; m->edge_log_steps = 0; m->edge_log_graphs = 0; m->layer_fallback = 0;
  m->tie_log = nullptr; m->tie_log_steps = 0; m->tie_log_graphs = 0; m->tie_tol = 0.f;
  // which fused-conv kernel the K=144 convs use: k_conv2 (persistent, one launch per layer, tail split) wins while a layer
  // is only a few rounds of workgroups (predict.py-sized batches), k_conv (conv.hip) at bench-sized batches.
  // DBFR_CONV2 = 0 / 1 forces one of them, default -1 = by batch size
  m->use_conv2 = getenv("DBFR_CONV2") ? atoi(getenv("DBFR_CONV2")) : -1;
  m->gemm_split = gemm_from_env();
  if (m->gemm_split < 0) rc = fail(DBFR_ERR_ARG, std::string("DBFR_GEMM=") + getenv("DBFR_GEMM") + ": unknown or retired value (f32 | split_f16 | reduce_first)");
  m->conv2_layers = getenv("DBFR_CONV2_LAYERS") ? atoi(getenv("DBFR_CONV2_LAYERS")) : 0;
  m->conv_fuse = getenv("DBFR_CONV_FUSE") ? atoi(getenv("DBFR_CONV_FUSE")) : 1;
  const char* fam[4] = {"lig_conv_layers", "cross_al_conv_layers", "atom_conv_layers", "cross_la_conv_layers"};
  for (int l = 0; l < cfg->num_conv_layers && !rc; ++l)
    for (int f = 0; f < 4 && !rc; ++f)
    {
      rc = pack_conv(m, tm, std::string(fam[f]) + "." + std::to_string(l), std::min(l, 3), &m->layer[l][f]);
      if (!rc) rc = pack_conv2(m, tm, std::string(fam[f]) + "." + std::to_string(l), std::min(l, 3), m->layer[l][f], &m->layer2[l][f]);
      if (!rc && m->layer2[l][f].f16_depth > f16_depth_ok()) { m->layer_fallback |= 1u << l; m->fallback_convs += std::string(fam[f]) + "." + std::to_string(l) + ";"; }
      if (!rc && (m->layer2[l][f].W2rinv || m->layer2[l][f].W1rinv)) m->rowscaled_convs += std::string(fam[f]) + "." + std::to_string(l) + ":" + std::to_string(m->layer2[l][f].f16_depth_run) + ";";
      if (!rc) rc = pack_conv2(m, tm, std::string(fam[f]) + "." + std::to_string(l), std::min(l, 3), m->layer[l][f], &m->layer2v[l][f], true);
      if (!rc) rc = pack_convz(m, tm, std::string(fam[f]) + "." + std::to_string(l), std::min(l, 3), m->layer2[l][f], &m->layerz[l][f]);
    }
  if (!rc) rc = pack_conv(m, tm, "final_conv", 4, &m->final_conv);
  if (!rc) rc = pack_conv(m, tm, "tor_bond_conv", 5, &m->tor_conv);
  if (!rc) rc = pack_conv2(m, tm, "tor_bond_conv", 5, m->tor_conv, &m->tor_conv2);
  if (!rc && m->tor_conv2.f16_depth > f16_depth_ok()) { m->layer_fallback |= 1u << 31; m->fallback_convs += "tor_bond_conv;"; }
  if (!rc && (m->tor_conv2.W2rinv || m->tor_conv2.W1rinv)) m->rowscaled_convs += "tor_bond_conv:" + std::to_string(m->tor_conv2.f16_depth_run) + ";";
  if (!rc) rc = pack_convz(m, tm, "tor_bond_conv", 5, m->tor_conv2, &m->tor_convz);
  if (!rc && !cfg->no_sc_torsion) rc = pack_conv(m, tm, "sc_tor_bond_conv", 5, &m->sc_conv);
  if (!rc && !cfg->no_sc_torsion) rc = pack_conv2(m, tm, "sc_tor_bond_conv", 5, m->sc_conv, &m->sc_conv2);
  if (!rc && !cfg->no_sc_torsion && m->sc_conv2.f16_depth > f16_depth_ok()) { m->layer_fallback |= 1u << 31; m->fallback_convs += "sc_tor_bond_conv;"; }
  if (!rc && !cfg->no_sc_torsion && (m->sc_conv2.W2rinv || m->sc_conv2.W1rinv)) m->rowscaled_convs += "sc_tor_bond_conv:" + std::to_string(m->sc_conv2.f16_depth_run) + ";";
  if (!rc && !cfg->no_sc_torsion) rc = pack_convz(m, tm, "sc_tor_bond_conv", 5, m->sc_conv2, &m->sc_convz);
  if (!rc) rc = pack_mlp(m, tm, "lig_node_embedding", cfg->lig_node_features + EMB, NS, NS, true, &m->lig_node_emb);
  if (!rc) rc = pack_mlp(m, tm, "lig_edge_embedding", cfg->lig_edge_features + 2 * EMB, NS, NS, true, &m->lig_edge_emb);
  if (!rc) rc = pack_mlp(m, tm, "atom_edge_embedding", 2 * EMB, NS, NS, true, &m->atom_edge_emb);
  if (!rc) rc = pack_mlp(m, tm, "la_edge_embedding", 2 * EMB, NS, NS, true, &m->la_edge_emb);
  if (!rc) rc = pack_mlp(m, tm, "center_edge_embedding", 2 * EMB, NS, NS, true, &m->center_edge_emb);
  if (!rc) rc = pack_mlp(m, tm, "tor_edge_embedding", EMB, NS, NS, true, &m->tor_edge_emb);
  if (!rc) rc = pack_mlp(m, tm, "tr_final_layer", 1 + EMB, NS, 1, true, &m->tr_final);
  if (!rc) rc = pack_mlp(m, tm, "rot_final_layer", 1 + EMB, NS, 1, true, &m->rot_final);
  if (!rc) rc = pack_mlp(m, tm, "tor_final_layer", 2 * NS, NS, 1, false, &m->tor_final);
  if (!rc && !cfg->no_sc_torsion) {
    rc = pack_mlp(m, tm, "sc_edge_embedding", EMB, NS, NS, true, &m->sc_edge_emb);
    if (!rc) rc = pack_mlp(m, tm, "sc_tor_final_layer", 2 * NS, NS, 1, false, &m->sc_final);
  }
  const int dims[5] = {37, 22, 4, 21, 2};
  for (int i = 0; i < 5 && !rc; ++i) {
    const float* e = need(tm, "atom_node_embedding.atom_emb_list." + std::to_string(i) + ".weight", (int64_t)dims[i] * NS, &rc);
    if (!rc) { m->atom_emb[i] = upload(m, std::vector<float>(e, e + dims[i] * NS), &rc); m->atom_dims[i] = dims[i]; }
  }
  if (!rc) {
    const float* w = need(tm, "atom_node_embedding.scalar_lin.weight", (int64_t)NS * (NS + EMB), &rc);
    if (!rc) {
      std::vector<float> t((size_t)(NS + EMB) * NS);
      for (int o = 0; o < NS; ++o) for (int i = 0; i < NS + EMB; ++i) t[(size_t)i * NS + o] = w[(size_t)o * (NS + EMB) + i];
      m->atom_lin_t = upload(m, t, &rc);
    }
  }
  struct { const char* n; const float** off; const float** c; } gs[4] = {
      {"lig", &m->gs_lig_off, &m->gs_lig_c}, {"atom", &m->gs_atom_off, &m->gs_atom_c},
      {"cross", &m->gs_cross_off, &m->gs_cross_c}, {"center", &m->gs_center_off, &m->gs_center_c}};
  for (int i = 0; i < 4 && !rc; ++i) {
    const float* off = need(tm, std::string(gs[i].n) + "_distance_expansion.offset", EMB, &rc);
    const float* c = need(tm, std::string(gs[i].n) + "_distance_expansion.coeff", 1, &rc);
    if (!rc) { *gs[i].off = upload(m, std::vector<float>(off, off + EMB), &rc); *gs[i].c = upload(m, std::vector<float>(c, c + 1), &rc); }
  }
  if (!rc) m->a14_group = upload(m, std::vector<int>(kAtom14ToGroup, kAtom14ToGroup + 21 * 14), &rc);
  // profiling counters: flops of the reference algorithm | reference-form bytes | fused-form bytes | executed flops | executed flops that are not padding | bytes of the form that runs
  if (!rc) { m->flops_dev = upload(m, std::vector<double>(6, 0.0), &rc); }
  if (!rc) { m->queue = upload(m, std::vector<int>(4, 0), &rc); }
  if (rc) { dbfr_model_destroy(m); return rc; }
  *out = m;
  return DBFR_OK;
}

extern "C" void dbfr_model_destroy(dbfr_model* m) {
  if (!m) return;
  for (void* p : m->allocs) (void)hipFree(p);
  for (auto e : m->ev) (void)hipEventDestroy(e);
  if (m->streams_ready) {
    for (int i = 0; i < 3; ++i) { (void)hipStreamDestroy(m->side[i]); (void)hipEventDestroy(m->ev_join[i]); }
    (void)hipEventDestroy(m->ev_fork); (void)hipEventDestroy(m->ev_a); (void)hipEventDestroy(m->ev_b);
  }
  delete m;
}

// ------------------------------------------------------------------------------------------------ workspace
struct WsEntry { std::string name; size_t offset, bytes; };
struct Bump {
  char* base; size_t off; size_t cap;
  std::vector<WsEntry>* log;
  template <typename T> T* take(size_t n, const char* name = nullptr) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    if (log && name) log->push_back({name, off, n * sizeof(T)});
    off += n * sizeof(T);
    return p;
  }
};

struct Ws {
  int* err; int64_t* counters;
  int *lig_batch, *atm_batch, *n_cab, *tor_batch, *sc_batch; uint8_t* is_cab;
  float* temb;
  float *c_t, *c_tr_sigma, *c_rot_norm, *c_tor_n2, *c_sc_n2;
  float *s_tr, *s_rot, *s_tor, *s_sc;
  float *lig_x[2], *atom_x[2];
  EdgeSet set[N_SETS];
  // centre set
  int *c_tgt, *c_gth, *c_row_start, *c_row_cnt, *c_n; float *c_dist, *c_sh, *c_emb;
  float* msg[4]; int multi; int conv2; float* gp; float *tor_attr, *sc_attr, *tor_feat, *sc_feat;
  float *xmax_l, *xmax_a;   // [G] largest |feature| of every graph's ligand / pocket nodes in the current layer (k_row_absmax; k_convz's y scale)
  int* n_edges6;  // [8] device counters
};

static void edge_set_take(Bump& b, EdgeSet& S, int cap, int n_targets, int G, int n_graphs, int* n_edges, const char* nm) {
  S.cap = cap; S.n_edges = n_edges;
  std::string p(nm);
  S.tgt = b.take<int>(cap, (p + ".tgt").c_str()); S.gth = b.take<int>(cap, (p + ".gth").c_str());
  S.aux = b.take<int>(cap, (p + ".aux").c_str());
  S.dist = b.take<float>(cap, (p + ".dist").c_str()); S.sh = b.take<float>((size_t)cap * SH_LD, (p + ".sh").c_str());
  S.emb = b.take<float>((size_t)cap * NS, (p + ".emb").c_str());
  S.row_start = b.take<int>(n_targets, (p + ".row_start").c_str()); S.row_cnt = b.take<int>(n_targets, (p + ".row_cnt").c_str());
  S.g_cnt = b.take<int>(G, (p + ".g_cnt").c_str()); S.g_base = b.take<int>(G, (p + ".g_base").c_str());
  S.chunk0 = b.take<int>(n_graphs + 1, (p + ".chunk0").c_str()); S.gedge0 = b.take<int>(n_graphs + 1, (p + ".gedge0").c_str());
  // k_convz's chunk table: a chunk ends after 32 edges, at the end of its graph or with the end of its CZ_MAXSEG-th target (graph.hip k_graph_chunks)
  S.chunk_cap = cap > 0 ? cap / 32 + n_targets / CZ_MAXSEG + n_graphs + 8 : 0;
  S.chunk_es = b.take<int>(std::max(S.chunk_cap, 1), (p + ".chunk_es").c_str()); S.chunk_gl = b.take<int>(std::max(S.chunk_cap, 1), (p + ".chunk_gl").c_str());
  S.seg_first = b.take<uint8_t>(std::max(cap, 1), (p + ".seg_first").c_str());
}

static int plan(const dbfr_model* m, const dbfr_batch* B, const dbfr_limits* lim, char* base, size_t cap, Ws* w,
                size_t* need_bytes, std::vector<WsEntry>* log = nullptr) {
  dbfr_limits L = {24, 64};
  if (lim) { if (lim->aa_avg_neighbors > 0) L.aa_avg_neighbors = lim->aa_avg_neighbors; if (lim->cross_avg_neighbors > 0) L.cross_avg_neighbors = lim->cross_avg_neighbors; }
  Bump b{base, 0, cap, log};
  const int G = B->G, NL = B->NL, NA = B->NA;
  w->err = b.take<int>(16, "err"); w->counters = (int64_t*)b.take<int64_t>(8); w->n_edges6 = b.take<int>(8, "n_edges");
  w->lig_batch = b.take<int>(NL); w->atm_batch = b.take<int>(NA); w->is_cab = b.take<uint8_t>(NA, "is_cab");
  w->n_cab = b.take<int>(G); w->tor_batch = b.take<int>(B->NTOR + 1); w->sc_batch = b.take<int>(B->NSC + 1);
  w->temb = b.take<float>((size_t)G * EMB, "temb");
  w->c_t = b.take<float>(G); w->c_tr_sigma = b.take<float>(G); w->c_rot_norm = b.take<float>(G);
  w->c_tor_n2 = b.take<float>(B->NTOR + 1); w->c_sc_n2 = b.take<float>(B->NSC + 1);
  w->s_tr = b.take<float>(3 * G); w->s_rot = b.take<float>(3 * G); w->s_tor = b.take<float>(B->NTOR + 1); w->s_sc = b.take<float>(B->NSC + 1);
  for (int i = 0; i < 2; ++i) { w->lig_x[i] = b.take<float>((size_t)NL * MAXD, i ? "lig_x1" : "lig_x0"); w->atom_x[i] = b.take<float>((size_t)NA * MAXD, i ? "atom_x1" : "atom_x0"); }
  const int lig_cap = m->cfg.lig_max_neighbors;
  const long cap_ll = (long)NL * std::min(lig_cap + 1, std::max(B->max_nl - 1, 1)) + B->EB;
  const long cap_aa = (long)NA * std::min(L.aa_avg_neighbors, std::max(B->max_na - 1, 1));
  const long cap_x = (long)NL * std::min(B->max_na, 2 * B->max_nr + L.cross_avg_neighbors);
  const long cap_t = (long)B->NTOR * lig_cap, cap_s = m->cfg.no_sc_torsion ? 0 : (long)B->NSC * lig_cap;
  const long caps[N_SETS] = {cap_ll, cap_aa, cap_x, cap_x, cap_t, cap_s};
  const int ntg[N_SETS] = {NL, NA, NL, NA, B->NTOR, B->NSC};
  long maxcap = NL;
  for (int k = 0; k < N_SETS; ++k) {
    if (caps[k] > 0x7fffff00L) return fail(DBFR_ERR_ARG, "batch too large for int32 edge indices; split it");
    static const char* names[N_SETS] = {"ll", "aa", "al", "la", "tor", "sc"};
    edge_set_take(b, w->set[k], (int)caps[k], ntg[k] + 1, G * dbfr_edge_chunks(B->max_na, B->max_nl), G, w->n_edges6 ? w->n_edges6 + k : nullptr, names[k]);
    maxcap = std::max(maxcap, caps[k]);
  }
  w->c_tgt = b.take<int>(NL); w->c_gth = b.take<int>(NL); w->c_dist = b.take<float>(NL); w->c_sh = b.take<float>((size_t)NL * SH_LD, "center.sh");
  w->c_emb = b.take<float>((size_t)NL * NS, "center.emb"); w->c_row_start = b.take<int>(G); w->c_row_cnt = b.take<int>(G);
  w->c_n = w->n_edges6 ? w->n_edges6 + 6 : nullptr;
  // Which conv path a call takes (measured on MI355X, tools/latency_run.py):
  //   largest edge set <= 128 k edges (predict.py-sized batches: 1 complex x 4 poses, -bs 16): persistent k_conv2, one launch per layer;
  //   above: k_conv_layer (the four convs of a layer as one k_conv grid) -- faster than k_conv2 from ~40 poses on and faster than
  //   the former four-streams-per-layer mode at every size (that mode is kept behind DBFR_MULTI_EDGES for comparison only).
  static const long multi_edges = getenv("DBFR_MULTI_EDGES") ? atol(getenv("DBFR_MULTI_EDGES")) : 0;
  static const long conv2_edges = getenv("DBFR_CONV2_EDGES") ? atol(getenv("DBFR_CONV2_EDGES")) : 128 * 1024;
  w->multi = maxcap <= multi_edges;
  w->conv2 = m->gemm_split || m->use_conv2 > 0 || (m->use_conv2 < 0 && maxcap <= conv2_edges);
  if (w->conv2 || m->conv2_layers > 0 || (m->conv_fuse && !w->multi)) {   // fused launches: every conv of a launch writes its own message buffer, sized by its own edge set
    // (layers left to k_conv in the mixed mode push all four convs through msg[0])
    const bool all_fused = w->conv2 || (m->conv_fuse && !w->multi && m->conv2_layers <= 0);
    const long mc[4] = {all_fused ? std::max({caps[SET_LL], (long)NL, cap_t, cap_s}) : maxcap, std::max(caps[SET_AL], cap_s), caps[SET_AA], caps[SET_LA]};
    for (int i = 0; i < 4; ++i) w->msg[i] = b.take<float>((size_t)std::max(mc[i], 1L) * MAXD, i == 0 ? "msg" : nullptr);
    if (w->conv2) w->multi = 0;
  } else
  for (int i = 0; i < 4; ++i) w->msg[i] = (i == 0 || w->multi) ? b.take<float>((size_t)maxcap * MAXD, i == 0 ? "msg" : nullptr) : nullptr;
  w->gp = b.take<float>((size_t)G * 12, "gp");
  w->xmax_l = b.take<float>(G, "xmax_l"); w->xmax_a = b.take<float>(G, "xmax_a");
  w->tor_attr = b.take<float>((size_t)(B->NTOR + 1) * NS, "tor_attr"); w->sc_attr = b.take<float>((size_t)(B->NSC + 1) * NS);
  w->tor_feat = b.take<float>((size_t)(B->NTOR + 1) * 2 * NS, "tor_feat"); w->sc_feat = b.take<float>((size_t)(B->NSC + 1) * 2 * NS);
  *need_bytes = b.off + 256;
  return DBFR_OK;
}

static int check_batch(const dbfr_model* m, const dbfr_batch* B) {
  if (!m || !B) return fail(DBFR_ERR_ARG, "null argument");
  if (B->G <= 0 || B->NL <= 0 || B->NA <= 0 || B->NR <= 0) return fail(DBFR_ERR_ARG, "empty batch");
  if (B->max_nl > 256) return fail(DBFR_ERR_ARG, "ligand with more than 256 heavy atoms is not supported");
  if (B->max_na > 8192) return fail(DBFR_ERR_ARG, "pocket with more than 8192 heavy atoms is not supported");
  if (B->max_nl <= 0 || B->max_na <= 0 || B->max_nr <= 0) return fail(DBFR_ERR_ARG, "max_nl/max_na/max_nr must be set");
  return DBFR_OK;
}

extern "C" int dbfr_workspace_bytes(const dbfr_model* m, const dbfr_batch* b, const dbfr_limits* lim, size_t* bytes) {
  int rc = check_batch(m, b);
  if (rc) return rc;
  if (!bytes) return fail(DBFR_ERR_ARG, "null argument");
  Ws w; memset(&w, 0, sizeof w);
  return plan(m, b, lim, nullptr, 0, &w, bytes);
}

// ------------------------------------------------------------------------------------------------ score network
// HIP-event pair around one fused-conv launch on its launch stream (profile mode 1)
static void prof_events(dbfr_model* m, hipEvent_t* e0, hipEvent_t* e1) {
  if (m->ev_used + 2 > m->ev.size()) {
    size_t old = m->ev.size();
    m->ev.resize(old + 512);
    for (size_t i = old; i < m->ev.size(); ++i) (void)hipEventCreate(&m->ev[i]);
  }
  *e0 = m->ev[m->ev_used++]; *e1 = m->ev[m->ev_used++];
}

static void conv_call(dbfr_model* m, const ConvW& cw, const int* n_edges, int max_edges, const int* tgt, const int* gth,
                      const float* emb, const float* sh, const float* tab1, int ld1, const int* idx1, const float* tab2,
                      int ld2, const int* idx2, const float* x, int ldx, float* msg, hipStream_t st) {
  ConvArgs a;
  a.n_edges = n_edges; a.max_edges = max_edges; a.tgt = tgt; a.gth = gth; a.emb = emb; a.sh = sh; a.sh_sign = 1.f;
  a.tab1 = tab1; a.ld1 = ld1; a.idx1 = idx1; a.tab2 = tab2; a.ld2 = ld2; a.idx2 = idx2; a.x = x; a.ldx = ldx;
  a.w = cw; a.msg = msg;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // the dominant kernel k_conv<144> only (k_conv<96>: 3 short launches per step).  profile 1: HIP events around every
  // launch (convs serialised on the main stream) + flop counter; profile 2: flop counter only, streams as in production
  const bool prof = m->profile == 1 && cw.K == 144;
  const bool count = m->profile && cw.K == 144;
  if (prof) {
    prof_events(m, &e0, &e1);
    (void)hipEventRecord(e0, st);
  }
  launch_conv(a, st);
  if (prof) (void)hipEventRecord(e1, st);
  if (count) {
    // algorithmic flops per edge: radial MLP 2K(K + W) + tensor-product contraction 2*(sum_paths mul1*mulo*dim_o)
    // second counter: HBM bytes the reference's two-kernel form moves per edge (the [E,W] weights once,
    // gathered irreps, harmonics, two int64 indices); the fused kernel never materialises them
    launch_acc_flops(n_edges, 2.0 * cw.K * ((double)cw.K + cw.W), 4.0 * (cw.W + cw.D_in + 9) + 16.0, fused_bytes(cw.D_in, cw.D_out), m->flops_dev, st);
  }
}

static Conv2Desc conv2_desc(const ConvW2& cw, const int* n_edges, int max_edges, const int* gth, const float* emb, const float* sh,
                            const float* tab1, int ld1, const int* idx1, const float* tab2, int ld2, const int* idx2, const float* x,
                            int ldx, float* msg) {
  Conv2Desc d;
  d.n_edges = n_edges; d.max_edges = max_edges; d.gth = gth; d.emb = emb; d.sh = sh; d.tab1 = tab1; d.ld1 = ld1; d.idx1 = idx1;
  d.tab2 = tab2; d.ld2 = ld2; d.idx2 = idx2; d.x = x; d.ldx = ldx; d.w = cw; d.msg = msg;
  return d;
}

static ConvZDesc convz_desc(const Conv2Desc& d, const ConvZ& z, const int* tgt, int D_out, const int* chunk_es, const int* chunk_gl, const int* n_chunks, int max_chunks,
                            const float* xmax = nullptr) {
  ConvZDesc o;
  o.n_edges = d.n_edges; o.max_edges = d.max_edges; o.tgt = tgt; o.gth = d.gth; o.emb = d.emb; o.sh = d.sh;
  o.tab1 = d.tab1; o.ld1 = d.ld1; o.idx1 = d.idx1; o.tab2 = d.tab2; o.ld2 = d.ld2; o.idx2 = d.idx2; o.x = d.x; o.ldx = d.ldx;
  o.w = z; o.msg = d.msg; o.D_out = D_out; o.chunk_es = chunk_es; o.chunk_gl = chunk_gl; o.n_chunks = n_chunks; o.max_chunks = max_chunks; o.xmax = xmax;
  return o;
}

// one fused k_conv2 launch over up to four K=144 convs (an interaction layer, or the two torsion heads)
// DBFR_GEMM_REDUCE_FIRST: `descs` hold the vector-output rows only (ConvW2 packed with vector_only) and `z` (same order) the reduce-first form of
// the scalar-output rows: k_conv2h for the former (convs without l = 1 outputs are left out), k_convz for the latter, both writing the same message buffers.
static void conv2_call(dbfr_model* m, const Conv2Desc* descs, const int* Ws, int n, hipStream_t st, bool f16_fallback = false, const ConvZDesc* z = nullptr) {
  Conv2Args a;
  memset(&a, 0, sizeof a);
  for (int i = 0; i < n; ++i) a.c[i] = descs[i];
  a.n_conv = n; a.queue = m->queue;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (m->profile == 1) {
    prof_events(m, &e0, &e1);
    (void)hipEventRecord(e0, st);
  }
  // developer clock read-out of k_conv2h: DBFR_CONV2_TRACE=<file>: the stamps of the LAST traced launch are written at exit
  static unsigned long long* trace_dev = nullptr;
  static const char* trace_file = getenv("DBFR_CONV2_TRACE");
  if (trace_file && !trace_dev) {
    if (hipMalloc(&trace_dev, 8 * C2_TRACE_CAP * sizeof(unsigned long long)) != hipSuccess) trace_dev = nullptr;
    else atexit([] {
      std::vector<unsigned long long> h(8 * C2_TRACE_CAP);
      if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
        if (FILE* f = fopen(getenv("DBFR_CONV2_TRACE"), "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
      }
    });
  }
  if (trace_dev) { (void)hipMemsetAsync(trace_dev, 0, 8 * C2_TRACE_CAP * sizeof(unsigned long long), st); a.trace = trace_dev; }
  if (z) {
    Conv2Args v = a;
    v.n_conv = 0;
    for (int i = 0; i < n; ++i)
      if (descs[i].w.n_tiles > 0) v.c[v.n_conv++] = descs[i];
    if (v.n_conv) launch_conv2h(v, st);
    ConvZArgs za;
    memset(&za, 0, sizeof za);
    for (int i = 0; i < n; ++i) za.c[i] = z[i];
    za.n_conv = n;
    za.dbg = nullptr;
    za.executed = m->profile ? m->flops_dev + 3 : nullptr;      // k_convz counts the matrix instructions it issues (one atomic per wave and unit)
    launch_convz(za, st);
  }
  else if (m->gemm_split >= DBFR_GEMM_SPLIT_F16 && !f16_fallback) launch_conv2h(a, st);
  else launch_conv2(a, st);   // DBFR_GEMM_F32, and a launch holding a conv that even per-row factors cannot fit into two fp16 pieces: the fp32 instruction
  if (m->profile == 1) (void)hipEventRecord(e1, st);
  if (m->profile) {
    AccBatch ab; ab.count = 0;
    auto acc = [&](const int* ne, double coef, double* dst) { if (ab.count < 32) { ab.n[ab.count] = ne; ab.coef[ab.count] = coef; ab.dst[ab.count] = dst; ++ab.count; } };
    for (int i = 0; i < n; ++i) {
      acc(descs[i].n_edges, 2.0 * 144 * (144.0 + Ws[i]), m->flops_dev);
      acc(descs[i].n_edges, 4.0 * (Ws[i] + descs[i].w.D_in + 9) + 16.0, m->flops_dev + 1);
      acc(descs[i].n_edges, fused_bytes(descs[i].w.D_in, descs[i].w.D_out), m->flops_dev + 2);
      // what the matrix pipe executes in the per-edge kernel of this launch: `products` MFMA flops per flop of the hidden layer and of the W2 rows it walks
      // (reduce-first: the vector-output rows only; the split kernels of round 2 keep the hidden layer on the fp32 instruction)
      const double rows = 16.0 * descs[i].w.n_tiles;
      const bool f16 = (m->gemm_split >= DBFR_GEMM_SPLIT_F16 && !f16_fallback) || z;
      const double ex = f16 ? 3.0 * 2.0 * 144 * (144.0 + rows) : 2.0 * 144 * (144.0 + rows);
      if (!z || descs[i].w.n_tiles > 0) acc(descs[i].n_edges, ex, m->flops_dev + 3);
      // ... of which are not padding: everything in the per-edge kernels, except that the pair of DBFR_GEMM_REDUCE_FIRST computes the hidden layer twice
      // (k_convz counts its own instructions and the useful ones among them itself, the hidden layer included)
      if (!z) acc(descs[i].n_edges, ex, m->flops_dev + 4);
      else if (descs[i].w.n_tiles > 0) acc(descs[i].n_edges, 3.0 * 2.0 * 144 * rows, m->flops_dev + 4);
      // HBM bytes of the form that RUNS: the fused form's, except that the pair reads the edge record and the gathered rows twice and writes the
      // scalar-output message columns once per SEGMENT (k_convz adds 4 x 48 per segment and irrep itself) and one flag byte per edge
      const int D_in = descs[i].w.D_in, D_out = descs[i].w.D_out;
      const double in_bytes = 4.0 * (48 + 9 + 3 + 48 + 48 + D_in);
      if (!z) acc(descs[i].n_edges, fused_bytes(D_in, D_out), m->flops_dev + 5);
      else acc(descs[i].n_edges, in_bytes + 1.0 + (descs[i].w.n_tiles > 0 ? in_bytes + 4.0 * (D_out - 48 * z[i].w.n_io) : 0.0), m->flops_dev + 5);
    }
    launch_acc_batch(ab, st);
  }
}

static int run_score(dbfr_model* m, const dbfr_batch* B, const dbfr_cond* c, const dbfr_scores* out, Ws& w,
                     hipStream_t st, int step = 0) {
  const dbfr_model_cfg& cfg = m->cfg;
  const int G = B->G, NL = B->NL, NA = B->NA;
  launch_time_embed(c->t, G, cfg.emb_scale, w.temb, st);
  // ---- graphs (all six radius-type sets in one count/scan/fill pass)
  GraphArgs ga;
  ga.b = *B; ga.lig_batch = w.lig_batch; ga.atm_batch = w.atm_batch; ga.is_cab = w.is_cab; ga.n_cab = w.n_cab;
  ga.tr_sigma = c->tr_sigma;
  ga.lig_cut2 = cfg.lig_cutoff * cfg.lig_cutoff; ga.atom_cut2 = cfg.atom_cutoff * cfg.atom_cutoff;
  ga.cross_cut2 = cfg.dynamic_max_cross ? 1.0f : cfg.cross_cutoff * cfg.cross_cutoff;
  ga.lig_cap = cfg.lig_max_neighbors; ga.atom_cap = cfg.atom_max_neighbors; ga.dynamic_cross = cfg.dynamic_max_cross;
  for (int k = 0; k < N_SETS; ++k) ga.set[k] = w.set[k];
  ga.err = w.err; ga.step = step; ga.lds_nl = ga.lds_na = 0;
  dbfr_edge_form(*B, &ga.n_chunk, &ga.lanes);      // one choice of the edge builder's form for every launcher of this call
  launch_edges(ga, false, st);
  // (a launcher that could not prepare its launch left the edge sets unbuilt: nothing below may run on them)
  if (int lrc = take_launch_error()) return lrc;
  if (m->gemm_split == DBFR_GEMM_REDUCE_FIRST) launch_graph_chunks(ga, st);   // per-graph chunks of 32 edges for k_convz
  // (the log rows are laid out for the graph capacity the caller sized the buffer for; a batch with more graphs was refused before anything was launched:
  // check_logs)
  if (m->edge_log && step < m->edge_log_steps) launch_edge_log(ga, m->edge_log + (size_t)step * N_SETS * m->edge_log_graphs, m->edge_log_graphs, st);
  if (m->tie_log && step < m->tie_log_steps) launch_edge_ties(ga, m->tie_log + (size_t)step * N_SETS * m->tie_log_graphs, m->tie_log_graphs, m->tie_tol, st);
  // ---- embeddings
  {
    MlpArgs a; memset(&a, 0, sizeof a);
    a.w = m->lig_node_emb; a.mode = IN_LIGNODE; a.n_rows_max = NL; a.temb = w.temb; a.row_graph_tab = w.lig_batch;
    a.lig_node = B->lig_node; a.nnode = cfg.lig_node_features; a.out = w.lig_x[0];
    launch_mlp(a, st);
  }
  {
    AtomEncArgs a; a.pocket_feat = B->pocket_feat; a.atm_batch = w.atm_batch; a.temb = w.temb;
    for (int i = 0; i < 5; ++i) { a.emb[i] = m->atom_emb[i]; a.dims[i] = m->atom_dims[i]; }
    a.lin_t = m->atom_lin_t; a.NA = NA; a.out = w.atom_x[0];
    launch_atom_encoder(a, st);
  }
  auto edge_mlp = [&](const Mlp2& mw, int mode, const EdgeSet& S, const int* tab, const float* off, const float* co) {
    MlpArgs a; memset(&a, 0, sizeof a);
    a.w = mw; a.mode = mode; a.n_rows_dev = S.n_edges; a.n_rows_max = S.cap; a.temb = w.temb; a.row_graph_tab = tab;
    a.tgt = S.tgt; a.aux = S.aux; a.dist = S.dist; a.bond_feat = B->bond_feat; a.nfeat = cfg.lig_edge_features;
    a.gs_offset = off; a.gs_coeff = co; a.out = S.emb;
    launch_mlp(a, st);
  };
  edge_mlp(m->lig_edge_emb, IN_LIGEDGE, w.set[SET_LL], w.lig_batch, m->gs_lig_off, m->gs_lig_c);
  edge_mlp(m->atom_edge_emb, IN_TG, w.set[SET_AA], w.atm_batch, m->gs_atom_off, m->gs_atom_c);
  edge_mlp(m->la_edge_emb, IN_TG, w.set[SET_AL], w.lig_batch, m->gs_cross_off, m->gs_cross_c);
  edge_mlp(m->la_edge_emb, IN_TG, w.set[SET_LA], w.atm_batch, m->gs_cross_off, m->gs_cross_c);
  // ---- interaction layers
  static const int dims[4] = {48, 84, 120, 168};
  int cur = 0;
  for (int l = 0; l < cfg.num_conv_layers; ++l) {
    const int Di = dims[std::min(l, 3)], Do = dims[std::min(l + 1, 3)];
    const float *lx = w.lig_x[cur], *ax = w.atom_x[cur];
    float *lnew = w.lig_x[cur ^ 1], *anew = w.atom_x[cur ^ 1];
    const EdgeSet &LL = w.set[SET_LL], &AA = w.set[SET_AA], &AL = w.set[SET_AL], &LA = w.set[SET_LA];
    if (w.conv2 || l < m->conv2_layers) {   // all four convs of the layer in ONE persistent launch (conv2.hip)
      const bool rf = m->gemm_split == DBFR_GEMM_REDUCE_FIRST && !((m->layer_fallback >> l) & 1u);
      const ConvW2* L2 = rf ? m->layer2v[l] : m->layer2[l];
      const Conv2Desc ds[4] = {
          conv2_desc(L2[0], LL.n_edges, LL.cap, LL.gth, LL.emb, LL.sh, lx, Di, LL.tgt, lx, Di, LL.gth, lx, Di, w.msg[0]),
          conv2_desc(L2[1], AL.n_edges, AL.cap, AL.gth, AL.emb, AL.sh, lx, Di, AL.tgt, ax, Di, AL.gth, ax, Di, w.msg[1]),
          conv2_desc(L2[2], AA.n_edges, AA.cap, AA.gth, AA.emb, AA.sh, ax, Di, AA.tgt, ax, Di, AA.gth, ax, Di, w.msg[2]),
          conv2_desc(L2[3], LA.n_edges, LA.cap, LA.gth, LA.emb, LA.sh, ax, Di, LA.tgt, lx, Di, LA.gth, lx, Di, w.msg[3])};
      const int Ws[4] = {m->layer[l][0].W, m->layer[l][1].W, m->layer[l][2].W, m->layer[l][3].W};
      if (rf) {   // the y scale of k_convz: largest |x| per graph and node set (x = the rows a conv gathers: LL, LA read ligand rows, AL, AA pocket rows)
        launch_row_absmax(lx, B->lig_ptr, w.xmax_l, ax, B->atm_ptr, w.xmax_a, Di, Di, G, NA, st);
      }
      auto zd = [&](int i, const EdgeSet& S, const float* xmax) { return convz_desc(ds[i], m->layerz[l][i], S.tgt, Do, S.chunk_es, S.chunk_gl, S.chunk0 + G, S.chunk_cap, xmax); };
      const ConvZDesc zs[4] = {zd(0, LL, w.xmax_l), zd(1, AL, w.xmax_a), zd(2, AA, w.xmax_a), zd(3, LA, w.xmax_l)};
      conv2_call(m, ds, Ws, 4, st, (m->layer_fallback >> l) & 1u, rf ? zs : nullptr);
      ReduceLayerArgs ra;
      const EdgeSet* es[4] = {&LL, &AL, &AA, &LA};
      for (int i = 0; i < 4; ++i) {
        ra.msg[i] = w.msg[i]; ra.row_start[i] = es[i]->row_start; ra.row_cnt[i] = es[i]->row_cnt; ra.ln[i] = m->layer[l][i].ln;
        ra.first[i] = rf ? es[i]->seg_first : nullptr; ra.sc_lanes[i] = rf ? convz_sc_lanes(m->layerz[l][i]) : 0;
      }
      ra.NL = NL; ra.NA = NA; ra.D = Do; ra.D_old = Di; ra.old_l = lx; ra.old_a = ax; ra.out_l = lnew; ra.out_a = anew;
      launch_reduce_ln_layer(ra, st);
    } else if (m->conv_fuse && !w.multi) {   // bench-sized batches: the layer's four convs as ONE k_conv grid, one reduction launch
      auto mk = [&](const ConvW& cw, const EdgeSet& S, const float* tab1, const int* idx1, const float* tab2, const int* idx2, const float* x, float* msg) {
        ConvArgs a;
        a.n_edges = S.n_edges; a.max_edges = S.cap; a.tgt = S.tgt; a.gth = S.gth; a.emb = S.emb; a.sh = S.sh; a.sh_sign = 1.f;
        a.tab1 = tab1; a.ld1 = Di; a.idx1 = idx1; a.tab2 = tab2; a.ld2 = Di; a.idx2 = idx2; a.x = x; a.ldx = Di; a.w = cw; a.msg = msg; a.trace = nullptr;
        return a;
      };
      const ConvArgs c4[4] = {mk(m->layer[l][0], LL, lx, LL.tgt, lx, LL.gth, lx, w.msg[0]), mk(m->layer[l][1], AL, lx, AL.tgt, ax, AL.gth, ax, w.msg[1]),
                              mk(m->layer[l][2], AA, ax, AA.tgt, ax, AA.gth, ax, w.msg[2]), mk(m->layer[l][3], LA, ax, LA.tgt, lx, LA.gth, lx, w.msg[3])};
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (m->profile == 1) {
        prof_events(m, &e0, &e1);
        (void)hipEventRecord(e0, st);
      }
      launch_conv_layer(c4, st);
      if (m->profile == 1) (void)hipEventRecord(e1, st);
      if (m->profile)
        for (int i = 0; i < 4; ++i)
          launch_acc_flops(c4[i].n_edges, 2.0 * 144 * (144.0 + m->layer[l][i].W), 4.0 * (m->layer[l][i].W + Di + 9) + 16.0, fused_bytes(Di, Do), m->flops_dev, st);
      ReduceLayerArgs ra;
      const EdgeSet* es[4] = {&LL, &AL, &AA, &LA};
      for (int i = 0; i < 4; ++i) { ra.msg[i] = w.msg[i]; ra.row_start[i] = es[i]->row_start; ra.row_cnt[i] = es[i]->row_cnt; ra.ln[i] = m->layer[l][i].ln; ra.first[i] = nullptr; ra.sc_lanes[i] = 0; }
      ra.NL = NL; ra.NA = NA; ra.D = Do; ra.D_old = Di; ra.old_l = lx; ra.old_a = ax; ra.out_l = lnew; ra.out_a = anew;
      launch_reduce_ln_layer(ra, st);
    } else if (!w.multi || m->profile == 1) {   // profiling times each conv alone on the main stream
      conv_call(m, m->layer[l][0], LL.n_edges, LL.cap, LL.tgt, LL.gth, LL.emb, LL.sh, lx, Di, LL.tgt, lx, Di, LL.gth, lx, Di, w.msg[0], st);
      launch_reduce_ln(w.msg[0], LL.row_start, LL.row_cnt, NL, Do, m->layer[l][0].ln, lx, Di, lnew, Do, 0, st);
      conv_call(m, m->layer[l][1], AL.n_edges, AL.cap, AL.tgt, AL.gth, AL.emb, AL.sh, lx, Di, AL.tgt, ax, Di, AL.gth, ax, Di, w.msg[0], st);
      launch_reduce_ln(w.msg[0], AL.row_start, AL.row_cnt, NL, Do, m->layer[l][1].ln, nullptr, 0, lnew, Do, 1, st);
      conv_call(m, m->layer[l][2], AA.n_edges, AA.cap, AA.tgt, AA.gth, AA.emb, AA.sh, ax, Di, AA.tgt, ax, Di, AA.gth, ax, Di, w.msg[0], st);
      launch_reduce_ln(w.msg[0], AA.row_start, AA.row_cnt, NA, Do, m->layer[l][2].ln, ax, Di, anew, Do, 0, st);
      conv_call(m, m->layer[l][3], LA.n_edges, LA.cap, LA.tgt, LA.gth, LA.emb, LA.sh, ax, Di, LA.tgt, lx, Di, LA.gth, lx, Di, w.msg[0], st);
      launch_reduce_ln(w.msg[0], LA.row_start, LA.row_cnt, NA, Do, m->layer[l][3].ln, nullptr, 0, anew, Do, 1, st);
    } else {
      // fork: all four convs read the OLD features; the two reductions into one node set stay ordered by an event
      hipStream_t s0 = m->side[0], s1 = m->side[1], s2 = m->side[2];
      HIPCHECK(hipEventRecord(m->ev_fork, st));
      for (int i = 0; i < 3; ++i) HIPCHECK(hipStreamWaitEvent(m->side[i], m->ev_fork, 0));
      conv_call(m, m->layer[l][0], LL.n_edges, LL.cap, LL.tgt, LL.gth, LL.emb, LL.sh, lx, Di, LL.tgt, lx, Di, LL.gth, lx, Di, w.msg[0], st);
      conv_call(m, m->layer[l][1], AL.n_edges, AL.cap, AL.tgt, AL.gth, AL.emb, AL.sh, lx, Di, AL.tgt, ax, Di, AL.gth, ax, Di, w.msg[1], s0);
      conv_call(m, m->layer[l][2], AA.n_edges, AA.cap, AA.tgt, AA.gth, AA.emb, AA.sh, ax, Di, AA.tgt, ax, Di, AA.gth, ax, Di, w.msg[2], s1);
      conv_call(m, m->layer[l][3], LA.n_edges, LA.cap, LA.tgt, LA.gth, LA.emb, LA.sh, ax, Di, LA.tgt, lx, Di, LA.gth, lx, Di, w.msg[3], s2);
      launch_reduce_ln(w.msg[0], LL.row_start, LL.row_cnt, NL, Do, m->layer[l][0].ln, lx, Di, lnew, Do, 0, st);
      HIPCHECK(hipEventRecord(m->ev_a, st));
      HIPCHECK(hipStreamWaitEvent(s0, m->ev_a, 0));
      launch_reduce_ln(w.msg[1], AL.row_start, AL.row_cnt, NL, Do, m->layer[l][1].ln, nullptr, 0, lnew, Do, 1, s0);
      launch_reduce_ln(w.msg[2], AA.row_start, AA.row_cnt, NA, Do, m->layer[l][2].ln, ax, Di, anew, Do, 0, s1);
      HIPCHECK(hipEventRecord(m->ev_b, s1));
      HIPCHECK(hipStreamWaitEvent(s2, m->ev_b, 0));
      launch_reduce_ln(w.msg[3], LA.row_start, LA.row_cnt, NA, Do, m->layer[l][3].ln, nullptr, 0, anew, Do, 1, s2);
      HIPCHECK(hipEventRecord(m->ev_join[0], s0));
      HIPCHECK(hipEventRecord(m->ev_join[2], s2));
      HIPCHECK(hipStreamWaitEvent(st, m->ev_join[0], 0));
      HIPCHECK(hipStreamWaitEvent(st, m->ev_join[2], 0));
    }
    cur ^= 1;
  }
  const int D = dims[std::min(cfg.num_conv_layers, 3)];
  if (D != MAXD) return fail(DBFR_ERR_ARG, "heads need num_conv_layers >= 3");
  const float *lx = w.lig_x[cur], *ax = w.atom_x[cur];
  // ---- the three heads are independent: in the small-batch (multi-stream) regime the two torsion heads run on side streams
  const bool fork_heads = w.multi && m->profile != 1 && !w.conv2;
  hipStream_t s_tor = fork_heads ? m->side[0] : st, s_sc = fork_heads ? m->side[1] : st;
  float* msg_tor = fork_heads ? w.msg[1] : w.msg[0];
  float* msg_sc = fork_heads ? w.msg[2] : w.msg[0];
  if (fork_heads) {
    HIPCHECK(hipEventRecord(m->ev_fork, st));
    HIPCHECK(hipStreamWaitEvent(s_tor, m->ev_fork, 0));
    HIPCHECK(hipStreamWaitEvent(s_sc, m->ev_fork, 0));
  }
  // ---- translation / rotation head
  launch_center_edges(*B, w.c_tgt, w.c_gth, w.c_dist, w.c_sh, w.c_row_start, w.c_row_cnt, st);
  {
    MlpArgs a; memset(&a, 0, sizeof a);
    a.w = m->center_edge_emb; a.mode = IN_TG; a.n_rows_max = NL; a.temb = w.temb; a.row_graph_tab = w.lig_batch;
    a.dist = w.c_dist; a.gs_offset = m->gs_center_off; a.gs_coeff = m->gs_center_c; a.out = w.c_emb;
    launch_mlp(a, st);
  }
  conv_call(m, m->final_conv, w.n_edges6 + 7, NL, w.c_tgt, w.c_gth, w.c_emb, w.c_sh, lx, D, w.c_gth, nullptr, 0, w.c_gth, lx, D, w.msg[0], st);
  launch_reduce_ln(w.msg[0], w.c_row_start, w.c_row_cnt, G, 12, m->final_conv.ln, nullptr, 0, w.gp, 12, 2, st);
  {
    TrRotArgs a; a.gp = w.gp; a.temb = w.temb; a.tr_sigma = c->tr_sigma; a.rot_norm = c->rot_score_norm;
    a.tr = m->tr_final; a.rot = m->rot_final; a.G = G; a.scale_by_sigma = cfg.scale_by_sigma; a.tr_out = out->tr;
    a.rot_out = out->rot; a.err = w.err;
    launch_trrot(a, st);
  }
  if (w.conv2) {   // both torsion heads: embeddings, ONE fused k_conv2 launch, reductions, final MLPs
    const EdgeSet& T = w.set[SET_TOR];
    const EdgeSet& S = w.set[SET_SC];
    const bool do_t = B->NTOR > 0, do_s = !cfg.no_sc_torsion && B->NSC > 0;
    const bool rf = m->gemm_split == DBFR_GEMM_REDUCE_FIRST && !((m->layer_fallback >> 31) & 1u);
    Conv2Desc ds[2]; ConvZDesc zs[2]; int Ws[2]; int nd = 0;
    if (rf) {
      launch_row_absmax(lx, B->lig_ptr, w.xmax_l, ax, B->atm_ptr, w.xmax_a, D, D, G, NA, st);
    }
    if (do_t) {
      launch_bond_attr(lx, D, B->bond_src, B->bond_dst, B->tor_bond, 0, B->NTOR, w.tor_attr, st);
      MlpArgs a; memset(&a, 0, sizeof a);
      a.w = m->tor_edge_emb; a.mode = IN_G; a.n_rows_dev = T.n_edges; a.n_rows_max = T.cap; a.dist = T.dist;
      a.gs_offset = m->gs_lig_off; a.gs_coeff = m->gs_lig_c; a.out = T.emb;
      launch_mlp(a, st);
      ds[nd] = conv2_desc(m->tor_conv2, T.n_edges, T.cap, T.gth, T.emb, T.sh, lx, D, T.gth, w.tor_attr, NS, T.tgt, lx, D, w.msg[0]);
      if (rf) { ds[nd].w.n_tiles = 0; zs[nd] = convz_desc(ds[nd], m->tor_convz, T.tgt, 2 * NS, T.chunk_es, T.chunk_gl, T.chunk0 + G, T.chunk_cap, w.xmax_l); }   // (all outputs of a torsion conv are scalars)
      Ws[nd++] = m->tor_conv.W;
    }
    if (do_s) {
      launch_bond_attr(ax, D, B->sc_bond, nullptr, nullptr, 2, B->NSC, w.sc_attr, st);
      MlpArgs a; memset(&a, 0, sizeof a);
      a.w = m->sc_edge_emb; a.mode = IN_G; a.n_rows_dev = S.n_edges; a.n_rows_max = S.cap; a.dist = S.dist;
      a.gs_offset = m->gs_atom_off; a.gs_coeff = m->gs_atom_c; a.out = S.emb;
      launch_mlp(a, st);
      ds[nd] = conv2_desc(m->sc_conv2, S.n_edges, S.cap, S.gth, S.emb, S.sh, ax, D, S.gth, w.sc_attr, NS, S.tgt, ax, D, w.msg[1]);
      if (rf) { ds[nd].w.n_tiles = 0; zs[nd] = convz_desc(ds[nd], m->sc_convz, S.tgt, 2 * NS, S.chunk_es, S.chunk_gl, S.chunk0 + G, S.chunk_cap, w.xmax_a); }
      Ws[nd++] = m->sc_conv.W;
    }
    if (nd) conv2_call(m, ds, Ws, nd, st, (m->layer_fallback >> 31) & 1u, rf ? zs : nullptr);
    if (do_t) {
      launch_reduce_ln(w.msg[0], T.row_start, T.row_cnt, B->NTOR, 2 * NS, m->tor_conv.ln, nullptr, 0, w.tor_feat, 2 * NS, 2, st, rf ? T.seg_first : nullptr, rf ? convz_sc_lanes(m->tor_convz) : 0);
      launch_tor_final(w.tor_feat, m->tor_final, c->tor_score_norm2, cfg.scale_by_sigma, B->NTOR, out->tor, st);
    }
    if (do_s) {
      launch_reduce_ln(w.msg[1], S.row_start, S.row_cnt, B->NSC, 2 * NS, m->sc_conv.ln, nullptr, 0, w.sc_feat, 2 * NS, 2, st, rf ? S.seg_first : nullptr, rf ? convz_sc_lanes(m->sc_convz) : 0);
      launch_tor_final(w.sc_feat, m->sc_final, c->sc_tor_score_norm2, cfg.scale_by_sigma, B->NSC, out->sc_tor, st);
    }
    return DBFR_OK;
  }
  // ---- ligand torsion head
  if (B->NTOR > 0) {
    const EdgeSet& T = w.set[SET_TOR];
    launch_bond_attr(lx, D, B->bond_src, B->bond_dst, B->tor_bond, 0, B->NTOR, w.tor_attr, s_tor);
    {
      MlpArgs a; memset(&a, 0, sizeof a);
      a.w = m->tor_edge_emb; a.mode = IN_G; a.n_rows_dev = T.n_edges; a.n_rows_max = T.cap; a.dist = T.dist;
      a.gs_offset = m->gs_lig_off; a.gs_coeff = m->gs_lig_c; a.out = T.emb;
      launch_mlp(a, s_tor);
    }
    conv_call(m, m->tor_conv, T.n_edges, T.cap, T.tgt, T.gth, T.emb, T.sh, lx, D, T.gth, w.tor_attr, NS, T.tgt, lx, D, msg_tor, s_tor);
    launch_reduce_ln(msg_tor, T.row_start, T.row_cnt, B->NTOR, 2 * NS, m->tor_conv.ln, nullptr, 0, w.tor_feat, 2 * NS, 2, s_tor);
    launch_tor_final(w.tor_feat, m->tor_final, c->tor_score_norm2, cfg.scale_by_sigma, B->NTOR, out->tor, s_tor);
  }
  // ---- side-chain torsion head
  if (!cfg.no_sc_torsion && B->NSC > 0) {
    const EdgeSet& S = w.set[SET_SC];
    launch_bond_attr(ax, D, B->sc_bond, nullptr, nullptr, 2, B->NSC, w.sc_attr, s_sc);
    {
      MlpArgs a; memset(&a, 0, sizeof a);
      a.w = m->sc_edge_emb; a.mode = IN_G; a.n_rows_dev = S.n_edges; a.n_rows_max = S.cap; a.dist = S.dist;
      a.gs_offset = m->gs_atom_off; a.gs_coeff = m->gs_atom_c; a.out = S.emb;
      launch_mlp(a, s_sc);
    }
    conv_call(m, m->sc_conv, S.n_edges, S.cap, S.tgt, S.gth, S.emb, S.sh, ax, D, S.gth, w.sc_attr, NS, S.tgt, ax, D, msg_sc, s_sc);
    launch_reduce_ln(msg_sc, S.row_start, S.row_cnt, B->NSC, 2 * NS, m->sc_conv.ln, nullptr, 0, w.sc_feat, 2 * NS, 2, s_sc);
    launch_tor_final(w.sc_feat, m->sc_final, c->sc_tor_score_norm2, cfg.scale_by_sigma, B->NSC, out->sc_tor, s_sc);
  }
  if (fork_heads) {
    HIPCHECK(hipEventRecord(m->ev_join[0], s_tor));
    HIPCHECK(hipEventRecord(m->ev_join[1], s_sc));
    HIPCHECK(hipStreamWaitEvent(st, m->ev_join[0], 0));
    HIPCHECK(hipStreamWaitEvent(st, m->ev_join[1], 0));
  }
  return DBFR_OK;
}

static int begin(dbfr_model* m, const dbfr_batch* B, void* workspace, size_t wbytes, const dbfr_limits* lim, Ws* w,
                 hipStream_t st) {
  g_launch_err = false;                                  // (a flag an earlier, failed call left on this thread is not this call's)
  int rc = check_batch(m, B);
  if (rc) return rc;
  rc = check_logs(m, B->G);
  if (rc) return rc;
  if (!workspace) return fail(DBFR_ERR_ARG, "null workspace");
  size_t needb = 0;
  memset(w, 0, sizeof *w);
  rc = plan(m, B, lim, (char*)workspace, wbytes, w, &needb);
  if (rc) return rc;
  if (needb > wbytes) return fail(DBFR_ERR_ARG, "workspace too small: need " + std::to_string(needb) + " bytes");
  if (w->multi && !m->streams_ready) {
    for (int i = 0; i < 3; ++i) {
      HIPCHECK(hipStreamCreateWithFlags(&m->side[i], hipStreamNonBlocking));
      HIPCHECK(hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming));
    }
    HIPCHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&m->ev_a, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&m->ev_b, hipEventDisableTiming));
    m->streams_ready = true;
  }
  HIPCHECK(hipMemsetAsync(workspace, 0, 1024, st));   // err @0, counters @256, n_edges6 @512
  if (w->conv2 || m->gemm_split || m->conv2_layers > 0) HIPCHECK(hipMemsetAsync(m->queue, 0, 16, st));   // the kernel re-arms it itself; this covers an aborted run
  launch_set_int(w->n_edges6 + 7, B->NL, st);           // the centre set has exactly one edge per ligand atom
  launch_batch_vectors(*B, w->lig_batch, w->atm_batch, w->is_cab, w->n_cab, w->tor_batch, w->sc_batch, st);
  return DBFR_OK;
}

extern "C" int dbfr_score(dbfr_model* m, const dbfr_batch* b, const dbfr_cond* cond, const dbfr_scores* out,
                          void* workspace, size_t workspace_bytes, const dbfr_limits* lim, void* hip_stream) {
  if (!cond || !out) return fail(DBFR_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)hip_stream;
  Ws w;
  int rc = begin(m, b, workspace, workspace_bytes, lim, &w, st);
  if (rc) return rc;
  rc = run_score(m, b, cond, out, w, st);
  if (rc) return rc;
  HIPCHECK(hipGetLastError());
  return take_launch_error();
}

extern "C" int dbfr_sample_range(dbfr_model* m, const dbfr_batch* b, const dbfr_step* steps, int32_t n_steps,
                                 int32_t step_begin, const dbfr_noise* noise, float* atom14_out, float* traj_lig,
                                 float* traj_atom14, void* workspace, size_t workspace_bytes, const dbfr_limits* lim,
                                 void* hip_stream) {
  if (!steps || n_steps <= 0 || !noise) return fail(DBFR_ERR_ARG, "null argument");
  if (step_begin < 0 || step_begin >= n_steps) return fail(DBFR_ERR_ARG, "step_begin out of range");
  hipStream_t st = (hipStream_t)hip_stream;
  Ws w;
  int rc = begin(m, b, workspace, workspace_bytes, lim, &w, st);
  if (rc) return rc;
  const int G = b->G;
  for (int s = step_begin; s < n_steps; ++s) {
    const dbfr_step& sp = steps[s];
    // set_time (scFlex.py:104-122): uniform conditioning over the batch
    launch_fill(w.c_t, sp.t, G, st);
    launch_fill(w.c_tr_sigma, sp.tr_sigma, G, st);
    launch_fill(w.c_rot_norm, sp.rot_score_norm, G, st);
    launch_fill(w.c_tor_n2, sp.tor_score_norm2, b->NTOR, st);
    launch_fill(w.c_sc_n2, sp.tor_score_norm2, b->NSC, st);
    dbfr_cond c = {w.c_t, w.c_tr_sigma, w.c_rot_norm, w.c_tor_n2, w.c_sc_n2};
    dbfr_scores sc = {w.s_tr, w.s_rot, w.s_tor, w.s_sc};
    rc = run_score(m, b, &c, &sc, w, st, s);
    if (rc) return rc;
    SdeLigArgs la;
    la.b = *b; la.tr_score = w.s_tr; la.rot_score = w.s_rot; la.tor_score = w.s_tor;
    la.z_tr = noise->z_tr + (size_t)s * G * 3; la.z_rot = noise->z_rot + (size_t)s * G * 3;
    la.z_tor = noise->z_tor + (size_t)s * b->NTOR;
    la.dt = sp.dt; la.tr_g2 = sp.tr_g2; la.tr_gsdt = sp.tr_gsdt; la.rot_g2 = sp.rot_g2; la.rot_gsdt = sp.rot_gsdt;
    la.tor_g2 = sp.tor_g2; la.tor_gsdt = sp.tor_gsdt;
    la.traj = traj_lig ? traj_lig + (size_t)s * b->NL * 3 : nullptr;
    la.err = w.err;
    launch_sde_ligand(la, st);
    if (!m->cfg.no_sc_torsion) {
      const bool last = s == n_steps - 1;
      launch_sidechain(*b, w.s_sc, noise->z_sc + (size_t)s * b->NSC, sp.dt, sp.sc_g2, sp.sc_gsdt, m->a14_group,
                       last ? atom14_out : nullptr, traj_atom14 ? traj_atom14 + (size_t)s * b->NR * 42 : nullptr, w.err, st);
    }
  }
  HIPCHECK(hipGetLastError());
  return take_launch_error();
}

extern "C" int dbfr_sample(dbfr_model* m, const dbfr_batch* b, const dbfr_step* steps, int32_t n_steps,
                           const dbfr_noise* noise, float* atom14_out, float* traj_lig, float* traj_atom14,
                           void* workspace, size_t workspace_bytes, const dbfr_limits* lim, void* hip_stream) {
  return dbfr_sample_range(m, b, steps, n_steps, 0, noise, atom14_out, traj_lig, traj_atom14, workspace, workspace_bytes, lim,
                           hip_stream);
}

extern "C" int dbfr_capacity_report(void* workspace, void* hip_stream, int32_t* first_failed_step, int64_t* needed_edges) {
  if (!workspace) return fail(DBFR_ERR_ARG, "null workspace");
  HIPCHECK(hipStreamSynchronize((hipStream_t)hip_stream));
  int h[16];
  HIPCHECK(hipMemcpy(h, workspace, sizeof h, hipMemcpyDeviceToHost));   // err block @0 (plan())
  if (first_failed_step) *first_failed_step = h[1] - 1;
  if (needed_edges) {   // dbfr_status_sync's counter order: lig, atom, cross(al), -, tor, sc_tor, cross(la), -
    needed_edges[0] = h[8 + SET_LL]; needed_edges[1] = h[8 + SET_AA]; needed_edges[2] = h[8 + SET_AL]; needed_edges[3] = 0;
    needed_edges[4] = h[8 + SET_TOR]; needed_edges[5] = h[8 + SET_SC]; needed_edges[6] = h[8 + SET_LA]; needed_edges[7] = 0;
  }
  return DBFR_OK;
}

extern "C" int dbfr_status_sync(void* workspace, void* hip_stream, int64_t* counters) {
  if (!workspace) return fail(DBFR_ERR_ARG, "null workspace");
  hipStream_t st = (hipStream_t)hip_stream;
  HIPCHECK(hipStreamSynchronize(st));
  int h[16 + 16 + 8];
  HIPCHECK(hipMemcpy(h, workspace, sizeof h, hipMemcpyDeviceToHost));
  // layout of plan(): err[16] @0 (64 B), counters[8] @256, n_edges6[8] @512
  int err = h[0];
  if (counters) {
    int ne[8];
    HIPCHECK(hipMemcpy(ne, (char*)workspace + 512, sizeof ne, hipMemcpyDeviceToHost));
    counters[0] = ne[SET_LL]; counters[1] = ne[SET_AA]; counters[2] = ne[SET_AL]; counters[3] = ne[7];
    counters[4] = ne[SET_TOR]; counters[5] = ne[SET_SC]; counters[6] = ne[SET_LA]; counters[7] = 0;
  }
  // device status word is a bit mask: 1 = an edge list overflowed its capacity, 2 = numeric check failed
  if (err & 1) return fail(DBFR_ERR_CAPACITY, "edge capacity exceeded: raise dbfr_limits");
  if (err & 2) return fail(DBFR_ERR_NUMERIC, "non-finite score or Kabsch determinant check failed");
  return DBFR_OK;
}

extern "C" int dbfr_init_poses(const dbfr_model* m, const dbfr_batch* b, const dbfr_init_tape* tape, float* atom14_out,
                               void* hip_stream) {
  int rc = check_batch(m, b);
  if (rc) return rc;
  if (!tape || !tape->rot || !tape->tr || !tape->sc_u || (b->NTOR > 0 && !tape->tor_u))
    return fail(DBFR_ERR_ARG, "null init tape");
  launch_init_poses(*b, *tape, m->a14_group, atom14_out, (hipStream_t)hip_stream);
  HIPCHECK(hipGetLastError());
  return DBFR_OK;
}

extern "C" int dbfr_extract_templates(int32_t n_res, const int32_t* aatype, const float* atom14_pos, float* backbone_transl,
                                      float* backbone_rots, float* default_frame, float* rigid_group_positions,
                                      float* torsion_angle, void* hip_stream) {
  if (n_res < 0 || !aatype || !atom14_pos || !backbone_transl || !backbone_rots || !default_frame || !rigid_group_positions ||
      !torsion_angle)
    return fail(DBFR_ERR_ARG, "null argument");
  launch_extract_templates(n_res, aatype, atom14_pos, backbone_transl, backbone_rots, default_frame, rigid_group_positions,
                           torsion_angle, (hipStream_t)hip_stream);
  HIPCHECK(hipGetLastError());
  return DBFR_OK;
}

extern "C" int dbfr_select_pocket(int32_t n_prot, int32_t n_res_total, const int32_t* res_ptr, int32_t atoms_per_res, const float* atom_pos,
                                  const float* atom_mask, const int32_t* ref_ptr, const float* ref_pos, double cutoff,
                                  int32_t max_neighbors, float* min_dist2, uint8_t* res_mask, void* hip_stream) {
  if (n_prot < 0 || n_res_total < 0 || atoms_per_res < 1) return fail(DBFR_ERR_ARG, "bad size");
  if (n_prot == 0 || n_res_total == 0) return DBFR_OK;
  if (!res_ptr || !atom_pos || !atom_mask || !ref_ptr || !ref_pos || !min_dist2 || !res_mask) return fail(DBFR_ERR_ARG, "null argument");
  // the reference compares float32 distances with the Python double cutoff ** 2 cast to float32
  launch_select_pocket(n_prot, n_res_total, res_ptr, atoms_per_res, atom_pos, atom_mask, ref_ptr, ref_pos, (float)(cutoff * cutoff),
                       max_neighbors, min_dist2, res_mask, (hipStream_t)hip_stream);
  HIPCHECK(hipGetLastError());
  return DBFR_OK;
}

extern "C" int dbfr_profile_enable(dbfr_model* m, int32_t on) {
  if (!m) return fail(DBFR_ERR_ARG, "null model");
  m->profile = on;
  return DBFR_OK;
}

extern "C" int dbfr_profile_read(dbfr_model* m, double* conv_ms, int64_t* conv_launches, double* conv_flops,
                                 double* ref_form_bytes, int32_t reset) {
  if (!m) return fail(DBFR_ERR_ARG, "null model");
  HIPCHECK(hipDeviceSynchronize());
  for (size_t i = 0; i + 1 < m->ev_used; i += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, m->ev[i], m->ev[i + 1]) == hipSuccess) { m->conv_ms_acc += ms; m->conv_launches_acc++; }
  }
  m->ev_used = 0;
  double fl[6] = {0, 0, 0, 0, 0, 0};
  HIPCHECK(hipMemcpy(fl, m->flops_dev, sizeof fl, hipMemcpyDeviceToHost));
  m->fused_bytes_last = fl[2];
  m->executed_last = fl[3];
  m->useful_last = fl[4];
  m->form_bytes_last = fl[5];
  if (conv_ms) *conv_ms = m->conv_ms_acc;
  if (conv_launches) *conv_launches = m->conv_launches_acc;
  if (conv_flops) *conv_flops = fl[0];
  if (ref_form_bytes) *ref_form_bytes = fl[1];
  if (reset) { m->conv_ms_acc = 0; m->conv_launches_acc = 0; HIPCHECK(hipMemset(m->flops_dev, 0, 6 * sizeof(double))); }
  return DBFR_OK;
}

extern "C" int dbfr_profile_executed_flops(const dbfr_model* m, double* executed_flops) {
  if (!m || !executed_flops) return fail(DBFR_ERR_ARG, "null argument");
  *executed_flops = m->executed_last;
  return DBFR_OK;
}

extern "C" int dbfr_profile_useful_flops(const dbfr_model* m, double* useful_flops, double* form_bytes) {
  if (!m || (!useful_flops && !form_bytes)) return fail(DBFR_ERR_ARG, "null argument");
  if (useful_flops) *useful_flops = m->useful_last;
  if (form_bytes) *form_bytes = m->form_bytes_last;
  return DBFR_OK;
}

extern "C" int dbfr_profile_fused_bytes(const dbfr_model* m, double* fused_form_bytes) {
  if (!m || !fused_form_bytes) return fail(DBFR_ERR_ARG, "null argument");
  *fused_form_bytes = m->fused_bytes_last;
  return DBFR_OK;
}

// test hook: named offsets of the internal buffers inside the workspace
extern "C" int dbfr_workspace_layout(const dbfr_model* m, const dbfr_batch* b, const dbfr_limits* lim, char* names,
                                     size_t names_cap, size_t* offsets, size_t* bytes, int32_t max_entries) {
  int rc = check_batch(m, b);
  if (rc) return rc;
  std::vector<WsEntry> log;
  Ws w; memset(&w, 0, sizeof w);
  size_t nb = 0;
  rc = plan(m, b, lim, nullptr, 0, &w, &nb, &log);
  if (rc) return rc;
  size_t pos = 0;
  int n = 0;
  for (auto& e : log) {
    if (n >= max_entries || pos + e.name.size() + 2 > names_cap) break;
    memcpy(names + pos, e.name.c_str(), e.name.size());
    pos += e.name.size();
    names[pos++] = ';';
    offsets[n] = e.offset; bytes[n] = e.bytes;
    ++n;
  }
  names[pos] = 0;
  return n;
}

// ------------------------------------------------------------------------------------------------ unit-test hooks
static const ConvW* pick_conv(const dbfr_model* m, int layer, int family) {
  if (layer >= 0 && layer < m->cfg.num_conv_layers && family >= 0 && family < 4) return &m->layer[layer][family];
  if (layer == -1) return &m->final_conv;
  if (layer == -2) return &m->tor_conv;
  if (layer == -3 && !m->cfg.no_sc_torsion) return &m->sc_conv;
  return nullptr;
}

static int test_conv_impl(dbfr_model* m, bool conv2, int32_t layer, int32_t family, int32_t n_edges, const int32_t* n_edges_dev,
                          const int32_t* tgt, const int32_t* gth, const float* emb, const float* sh, const float* tab1, int32_t ld1,
                          const int32_t* idx1, const float* tab2, int32_t ld2, const int32_t* idx2, const float* x, int32_t ldx,
                          float* msg, void* hip_stream) {
  if (!m) return fail(DBFR_ERR_ARG, "null model");
  g_launch_err = false;
  const ConvW* cw = pick_conv(m, layer, family);
  if (!cw) return fail(DBFR_ERR_ARG, "no such conv");
  if (conv2) {
    if (cw->K != 144) return fail(DBFR_ERR_ARG, "k_conv2 serves the K=144 convs");
    const ConvW2* cw2 = layer >= 0 ? &m->layer2[layer][family] : layer == -2 ? &m->tor_conv2 : &m->sc_conv2;
    const bool deep = cw2->f16_depth > f16_depth_ok();
    const bool rf = m->gemm_split == DBFR_GEMM_REDUCE_FIRST && !deep;   // (the message buffer then holds segment sums in the segments' first rows: include/dbfr.h)
    Conv2Desc d = conv2_desc(rf && layer >= 0 ? m->layer2v[layer][family] : *cw2, n_edges_dev, n_edges, gth, emb, sh, tab1, ld1, idx1, tab2, ld2, idx2, x, ldx, msg);
    if (rf && layer < 0) d.w.n_tiles = 0;
    // k_convz's chunk table for this flat edge list, in a scratch buffer the hook keeps (test hook: one caller at a time): the list is cut every
    // 2048 edges as if those were graphs (parallel walk)
    static int* scratch_dev[16] = {nullptr}; static size_t scratch_dev_ints[16] = {0};   // (per device: the hook may be driven on several)
    int dev_id = 0;
    HIPCHECK(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= 16) return fail(DBFR_ERR_ARG, "test hook: device index beyond 15");
    int*& scratch = scratch_dev[dev_id]; size_t& scratch_ints = scratch_dev_ints[dev_id];
    const int span = 2048, n_span = std::max((n_edges + span - 1) / span, 1), ccap = n_edges / 32 + n_edges / CZ_MAXSEG + n_span + 8;
    if (rf) {
      const size_t need = (size_t)n_span + 1 + 2 * (size_t)ccap;
      if (need > scratch_ints) {
        if (scratch) HIPCHECK(hipFree(scratch));           // (hipFree waits for the device: nothing in flight still reads the old buffer)
        scratch = nullptr; scratch_ints = 0;
        HIPCHECK(hipMalloc(&scratch, need * sizeof(int)));
        scratch_ints = need;
      }
      launch_flat_chunks(tgt, n_edges_dev, n_edges, span, n_span, scratch, ccap, scratch + n_span + 1, scratch + n_span + 1 + ccap, (hipStream_t)hip_stream);
    }
    const ConvZDesc z = convz_desc(d, layer >= 0 ? m->layerz[layer][family] : layer == -2 ? m->tor_convz : m->sc_convz, tgt, cw->D_out,
                                   scratch ? scratch + n_span + 1 : nullptr, scratch ? scratch + n_span + 1 + ccap : nullptr, scratch ? scratch + n_span : nullptr, ccap);
    const int W = cw->W;
    // (k_convz writes the scalar-output columns of a segment's first row only; the hook's documented layout has zeros in the other rows)
    if (rf) HIPCHECK(hipMemsetAsync(msg, 0, (size_t)n_edges * cw->D_out * sizeof(float), (hipStream_t)hip_stream));
    conv2_call(m, &d, &W, 1, (hipStream_t)hip_stream, deep, rf ? &z : nullptr);
  } else {
    conv_call(m, *cw, n_edges_dev, n_edges, tgt, gth, emb, sh, tab1, ld1, idx1, tab2, ld2, idx2, x, ldx, msg, (hipStream_t)hip_stream);
  }
  HIPCHECK(hipGetLastError());
  return take_launch_error();
}

// Test hook: the chunk table of DBFR_GEMM_REDUCE_FIRST for one flat, target-sorted edge list, cut every `span` edges as if those were graphs
// (graph.hip k_flat_chunk_*; the same chunk_len as the per-graph kernels of the sampler).  Device pointers; scratch = n_span + 1 ints.
extern "C" int dbfr_test_chunk_table(const int32_t* tgt, const int32_t* n_edges_dev, int32_t max_edges, int32_t span, int32_t* scratch, int32_t cap,
                                     int32_t* chunk_es, int32_t* chunk_gl, void* hip_stream) {
  if (!tgt || !n_edges_dev || !scratch || !chunk_es || !chunk_gl || span <= 0 || max_edges < 0 || cap <= 0) return fail(DBFR_ERR_ARG, "dbfr_test_chunk_table: bad argument");
  g_launch_err = false;
  const int n_span = std::max((max_edges + span - 1) / span, 1);
  launch_flat_chunks(tgt, n_edges_dev, max_edges, span, n_span, scratch, cap, chunk_es, chunk_gl, (hipStream_t)hip_stream);
  HIPCHECK(hipGetLastError());
  return take_launch_error();
}

extern "C" int dbfr_test_conv(dbfr_model* m, int32_t layer, int32_t family, int32_t n_edges, const int32_t* n_edges_dev,
                              const int32_t* tgt, const int32_t* gth, const float* emb, const float* sh,
                              const float* tab1, int32_t ld1, const int32_t* idx1, const float* tab2, int32_t ld2,
                              const int32_t* idx2, const float* x, int32_t ldx, float* msg, void* hip_stream) {
  return test_conv_impl(m, false, layer, family, n_edges, n_edges_dev, tgt, gth, emb, sh, tab1, ld1, idx1, tab2, ld2, idx2, x, ldx,
                        msg, hip_stream);
}

extern "C" int dbfr_test_conv2(dbfr_model* m, int32_t layer, int32_t family, int32_t n_edges, const int32_t* n_edges_dev,
                               const int32_t* tgt, const int32_t* gth, const float* emb, const float* sh,
                               const float* tab1, int32_t ld1, const int32_t* idx1, const float* tab2, int32_t ld2,
                               const int32_t* idx2, const float* x, int32_t ldx, float* msg, void* hip_stream) {
  return test_conv_impl(m, true, layer, family, n_edges, n_edges_dev, tgt, gth, emb, sh, tab1, ld1, idx1, tab2, ld2, idx2, x, ldx,
                        msg, hip_stream);
}

// (ABI 6) ... with the message interface of DBFR_GEMM_REDUCE_FIRST: seg_first[e] = 1 marks the rows whose scalar-output columns hold a segment's sum (the
// other rows' scalar columns are not read -- they may hold anything), the vector columns are per edge; seg_first == NULL: every column of every row
extern "C" int dbfr_test_reduce_ln2(dbfr_model* m, int32_t layer, int32_t family, const float* msg, const int32_t* row_start, const int32_t* row_cnt,
                                    int32_t n_nodes, const float* old, int32_t d_old, float* out, int32_t mode, const uint8_t* seg_first, void* hip_stream) {
  if (!m) return fail(DBFR_ERR_ARG, "null model");
  const ConvW* cw = pick_conv(m, layer, family);
  if (!cw) return fail(DBFR_ERR_ARG, "no such conv");
  unsigned long long lanes = 0;
  if (seg_first) {
    if (layer == -1 || cw->K != 144) return fail(DBFR_ERR_ARG, "dbfr_test_reduce_ln2: seg_first with a conv that has no reduce-first form");
    lanes = convz_sc_lanes(layer >= 0 ? m->layerz[layer][family] : layer == -2 ? m->tor_convz : m->sc_convz);
  }
  launch_reduce_ln(msg, row_start, row_cnt, n_nodes, cw->D_out, cw->ln, old, d_old, out, cw->D_out, mode, (hipStream_t)hip_stream, seg_first, lanes);
  HIPCHECK(hipGetLastError());
  return DBFR_OK;
}

extern "C" int dbfr_test_reduce_ln(dbfr_model* m, int32_t layer, int32_t family, const float* msg,
                                   const int32_t* row_start, const int32_t* row_cnt, int32_t n_nodes, const float* old,
                                   int32_t d_old, float* out, int32_t mode, void* hip_stream) {
  if (!m) return fail(DBFR_ERR_ARG, "null model");
  const ConvW* cw = pick_conv(m, layer, family);
  if (!cw) return fail(DBFR_ERR_ARG, "no such conv");
  launch_reduce_ln(msg, row_start, row_cnt, n_nodes, cw->D_out, cw->ln, old, d_old, out, cw->D_out, mode,
                   (hipStream_t)hip_stream);
  HIPCHECK(hipGetLastError());
  return DBFR_OK;
}
