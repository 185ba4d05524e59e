/*
 * dbfr.h -- C ABI of the MI355X-native DiffBindFR reverse-diffusion sampler.
 *
 * The reference (HBioquant/DiffBindFR) is 100% Python: it has no FFI for this
 * path.  Its operator boundary is the mmcv-style registry + nn.Module contract
 *   INTERACTION['TensorProductModel']   druglib/models/Docking/interaction/tpscore.py:202-573
 *   MLDOCK_BUILDER['DiffBindFR']        druglib/models/Docking/scFlex.py:26-250
 * (druglib/models/builder.py:7-34).  This header is the C ABI underneath the
 * Python classes that plug into those registries (diffbindfr_amd/score_model.py,
 * diffbindfr_amd/sampler.py); INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / hip types in signatures
 *     (a hipStream_t travels as void*; NULL = the null stream);
 *   - every entry point returns 0 on success or a negative dbfr_status;
 *     dbfr_last_error() gives the message of the last failure on this thread;
 *   - "host" pointers are read during the call and not retained; "device"
 *     pointers must stay valid until the stream work that uses them is done;
 *   - one model handle per device; a handle is thread-compatible, not thread-safe;
 *   - all arithmetic is fp32 (the reference runs fp32: base.py:22 fp16 off);
 *     indices are int32 on the device side.
 */
#ifndef DBFR_H
#define DBFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DBFR_ABI_VERSION 6   /* 2: + dbfr_sample_range, dbfr_capacity_report, dbfr_sdf_*, dbfr_mdn_*, dbfr_build_id, dbfr_test_conv2;
                                3: + dbfr_model_set_edge_log, dbfr_model_fallback_convs, dbfr_test_pack_f16_depth, dbfr_probe_mfma_f16 (additions only);
                                4: + DBFR_GEMM_REDUCE_FIRST (the new default), dbfr_profile_executed_flops; dbfr_model_set_edge_log takes the graph capacity; DBFR_GEMM_SPLIT_BF16_L1 (k_conv2s) retired; dbfr_test_conv2's message rows in that mode hold segment sums;
                                5: + dbfr_model_rowscaled_convs (per-row factors instead of the three-bf16-piece fall-back), dbfr_test_pack_f16_rows, dbfr_test_chunk_table; the reduce-first chunks hold <= 4 targets; DBFR_GEMM_SPLIT_BF16 (k_conv2r) retired;
                                6: + dbfr_profile_useful_flops, dbfr_model_set_tie_log, dbfr_test_reduce_ln2; dbfr_model_set_edge_log accepts batches with fewer graphs than its capacity; an unknown DBFR_GEMM value fails dbfr_model_create */

typedef enum {
  DBFR_OK = 0,
  DBFR_ERR_ARG = -1,       /* bad argument / unsupported configuration            */
  DBFR_ERR_HIP = -2,       /* a HIP runtime call failed                           */
  DBFR_ERR_CAPACITY = -3,  /* workspace edge capacity exceeded (see dbfr_limits)  */
  DBFR_ERR_SELFTEST = -4,  /* built-in Clebsch-Gordan closed forms failed self-check */
  DBFR_ERR_NUMERIC = -5    /* non-finite score / Kabsch determinant check failed  */
} dbfr_status;

typedef struct dbfr_model dbfr_model; /* packed weights resident in HBM */

/* ---- model hyper-parameters (DiffBindFR/configs/diffbindfr_ts.py:107-142).
 * Only the reference's inference configuration family is supported:
 * ns=48, nv=12, sh_lmax=2, 32-d embeddings, use_second_order_repr=False.      */
typedef struct {
  int32_t ns, nv, sh_lmax;
  int32_t num_conv_layers;
  int32_t lig_node_features, lig_edge_features;
  int32_t distance_embed_dim, sigma_embed_dim;
  float   emb_scale;
  float   lig_cutoff, atom_cutoff, cross_cutoff, center_max_distance;
  int32_t atom_max_neighbors;  /* radius_graph cap for the pocket graph (1000)   */
  int32_t lig_max_neighbors;   /* torch_cluster default cap (32), tpscore.py:586 */
  int32_t dynamic_max_cross;   /* 1: cross cutoff = 0.2*tr_sigma + 5             */
  int32_t scale_by_sigma;
  int32_t no_sc_torsion;
} dbfr_model_cfg;

/* One named fp32 tensor of the reference state_dict (host memory, row-major,
 * exactly the shape the reference module holds; names relative to
 * TensorProductModel, e.g. "lig_conv_layers.3.fc.lin.3.weight").               */
typedef struct {
  const char*  name;
  const float* data;
  int64_t      numel;
} dbfr_tensor;

/* Builds the device-resident packed model (radial-MLP weights re-tiled into
 * MFMA fragment order with the tensor-product path normalisation folded in).
 * Replaces: TensorProductModel.__init__ + load_checkpoint(strict=True)
 * (tpscore.py:215-410; DiffBindFR/common/engines.py:137-165).  Every tensor of
 * SURVEY.md Appendix B.3 must be present; unknown names are an error.          */
int dbfr_model_create(const dbfr_model_cfg* cfg, const dbfr_tensor* tensors, int32_t n_tensors,
                      dbfr_model** out);
void dbfr_model_destroy(dbfr_model* m);

/* ---- one collated batch of G graphs (complex x pose), device pointers.
 * Layout = the reference's batched dict with int32
 * indices, CSR pointers instead of batch vectors, and the ragged python lists
 * flattened.  Tensors marked in/out are advanced in place by dbfr_sample.       */
typedef struct {
  int32_t G;        /* graphs                                                    */
  int32_t NL;       /* ligand heavy atoms, all graphs                            */
  int32_t NA;       /* pocket heavy atoms, all graphs                            */
  int32_t NR;       /* pocket residues, all graphs                               */
  int32_t EB;       /* directed ligand bonds (both directions), sorted by src    */
  int32_t NTOR;     /* ligand rotatable bonds (tor_edge_mask.sum())              */
  int32_t NSC;      /* side-chain torsions (sc_torsion_edge_mask.sum())          */
  int32_t max_nl, max_na, max_nr; /* per-graph maxima (host-known)               */
  /* ligand */
  const int32_t* lig_ptr;     /* [G+1]                                           */
  const float*   lig_node;    /* [NL, lig_node_features]                         */
  float*         lig_pos;     /* [NL,3]                               in/out     */
  const int32_t* bond_src;    /* [EB] global ligand atom index                   */
  const int32_t* bond_dst;    /* [EB]                                            */
  const float*   bond_feat;   /* [EB, lig_edge_features]                         */
  const int32_t* bond_ptr;    /* [NL+1] CSR of bonds by bond_src                 */
  const int32_t* tor_ptr;     /* [G+1]  torsions per graph                       */
  const int32_t* tor_bond;    /* [NTOR] index into bond_* (masked bonds, in order) */
  const uint8_t* rot_mask;    /* rot_node_mask rows, one byte per ligand atom    */
  const int64_t* rot_mask_off;/* [NTOR] byte offset of each row in rot_mask      */
  /* pocket */
  const int32_t* atm_ptr;     /* [G+1]                                           */
  const int32_t* res_ptr;     /* [G+1]                                           */
  const float*   pocket_feat; /* [NA,5] (atom37, coarse22, element4, aatype, is_backbone) */
  float*         rec_pos;     /* [NA,3]                               in/out     */
  const int32_t* sequence;    /* [NR]                                            */
  const float*   backbone_transl; /* [NR,3]                                      */
  const float*   backbone_rots;   /* [NR,3,3]                                    */
  const float*   default_frame;   /* [NR,8,4,4]                                  */
  const float*   rigid_group_positions; /* [NR,14,3]                             */
  float*         torsion_angle;   /* [NR,5] (psi, chi1..4) radians    in/out     */
  const int32_t* atom14_slot;     /* [NR,14] compact atom index or -1 (atom14_mask) */
  const int32_t* sc_res_chi;      /* [NSC] res*4+k of each masked chi, row-major */
  const int32_t* sc_bond;         /* [NSC,2] global atom ids (j,k) of the chi bond */
  const int32_t* sc_ptr;          /* [G+1] side-chain torsions per graph         */
} dbfr_batch;

/* Capacity knobs of the per-step edge lists (graphs are rebuilt every step).    */
typedef struct {
  int32_t aa_avg_neighbors;     /* pocket graph: mean edges per atom budgeted (default 24)   */
  int32_t cross_avg_neighbors;  /* non-CA/CB pocket atoms per ligand atom budgeted (default 64) */
} dbfr_limits;

/* Bytes of device workspace dbfr_score/dbfr_sample need for this batch shape.   */
int dbfr_workspace_bytes(const dbfr_model* m, const dbfr_batch* b, const dbfr_limits* lim, size_t* bytes);

/* ---- level 1: one score-network evaluation.
 * Replaces TensorProductModel.forward (tpscore.py:462-573).  Per-graph /
 * per-torsion conditioning arrives as device arrays exactly as set_time
 * produces it (scFlex.py:104-122).  Outputs are device arrays.                  */
typedef struct {
  const float* t;                 /* [G]                                          */
  const float* tr_sigma;          /* [G]                                          */
  const float* rot_score_norm;    /* [G]                                          */
  const float* tor_score_norm2;   /* [NTOR]                                       */
  const float* sc_tor_score_norm2;/* [NSC] (already masked/compacted)             */
} dbfr_cond;

typedef struct {
  float* tr;      /* [G,3]  */
  float* rot;     /* [G,3]  */
  float* tor;     /* [NTOR] */
  float* sc_tor;  /* [NSC]  */
} dbfr_scores;

int dbfr_score(dbfr_model* m, const dbfr_batch* b, const dbfr_cond* cond, const dbfr_scores* out,
               void* workspace, size_t workspace_bytes, const dbfr_limits* lim, void* hip_stream);

/* ---- level 2: the whole reverse-SDE sampler.
 * Replaces DiffBindFR.sample (scFlex.py:124-250): `n_steps` Euler-Maruyama steps
 * { set_time -> score net -> perturbations -> ligand rigid+torsion update + Kabsch
 *   -> chi update + side-chain rebuild }, entirely on the device.
 * Per-step scalars come from the host schedule (diffbindfr_amd/schedule.py mirrors
 * scFlex.py:83-122,154-161); the N(0,1) tape is supplied by the caller (device).  */
typedef struct {
  float t, dt;
  float tr_sigma, rot_score_norm, tor_score_norm2;   /* conditioning (uniform over graphs) */
  float tr_g2, tr_gsdt;      /* g^2 and g*sqrt(dt) as the reference's fp32 ops yield them */
  float rot_g2, rot_gsdt;
  float tor_g2, tor_gsdt;
  float sc_g2, sc_gsdt;
} dbfr_step;

typedef struct {
  const float* z_tr;   /* [n_steps, G, 3]  */
  const float* z_rot;  /* [n_steps, G, 3]  */
  const float* z_tor;  /* [n_steps, NTOR]  */
  const float* z_sc;   /* [n_steps, NSC]   */
} dbfr_noise;

/* atom14_out: [NR,14,3] device, masked atom14 positions after the last step
 * (may be NULL).  traj_lig / traj_atom14: optional [n_steps,...] device buffers
 * receiving every step (visualize=True), else NULL.                              */
int dbfr_sample(dbfr_model* m, const dbfr_batch* b, const dbfr_step* steps, int32_t n_steps,
                const dbfr_noise* noise, float* atom14_out, float* traj_lig, float* traj_atom14,
                void* workspace, size_t workspace_bytes, const dbfr_limits* lim, void* hip_stream);

/* The same, steps [step_begin, n_steps) only: `steps`, `noise` and the trajectory buffers are indexed by the absolute
 * step.  Used to RESUME after DBFR_ERR_CAPACITY: once an edge list of step s does not fit, that set is left empty,
 * the pose updates of step s and of every later step are skipped (the state stays as it was at the beginning of step
 * s) and the device remembers s and the edge counts it needed.  The caller reads them with dbfr_capacity_report,
 * raises dbfr_limits, sizes a new workspace and calls dbfr_sample_range(step_begin = s).                           */
int dbfr_sample_range(dbfr_model* m, const dbfr_batch* b, const dbfr_step* steps, int32_t n_steps, int32_t step_begin,
                      const dbfr_noise* noise, float* atom14_out, float* traj_lig, float* traj_atom14,
                      void* workspace, size_t workspace_bytes, const dbfr_limits* lim, void* hip_stream);

/* After dbfr_status_sync returned DBFR_ERR_CAPACITY: the first step whose edge lists overflowed (-1: none) and the
 * largest edge count each set needed so far, [8] int64 in dbfr_status_sync's counter order
 * {lig, atom, cross lig<-atom, 0, tor, sc_tor, cross atom<-lig, 0}.  Synchronises the stream.                      */
int dbfr_capacity_report(void* workspace, void* hip_stream, int32_t* first_failed_step, int64_t* needed_edges);

/* Per-graph read-out of the per-step graphs (dbfr_status_sync's counters are per batch).  With log != NULL every step s < n_steps_cap of
 * the dbfr_sample / dbfr_sample_range calls that follow on this model writes the edge count of graph g in edge set k to
 * log[(s * 6 + k) * n_graphs_cap + g] (device int32, caller-owned, n_steps_cap * 6 * n_graphs_cap entries; a batch with MORE graphs than
 * n_graphs_cap is refused with DBFR_ERR_ARG before anything is launched, a smaller one -- the ragged last batch of a sharded run -- fills the
 * first G entries of each row; until ABI 6 any other count was refused, and only after the step's first launches), k = {0 ligand
 * (bonds + radius_graph, tpscore.py:586), 1 pocket (:613), 2 cross lig<-atom, 3 cross atom<-lig (the same pairs, :655-660), 4 ligand
 * torsion (:721), 5 side-chain torsion (:747)}; dbfr_score writes row s = 0.  A set that overflowed its capacity still reports the count
 * it needed.  log == NULL switches the read-out off (the default).  The counts make a hard-cutoff event visible: two runs whose
 * coordinates differ in the 5th decimal build different graphs exactly where a pair sits within rounding distance of a cutoff
 * (tests/test_examples.py).                                                                                                        */
int dbfr_model_set_edge_log(dbfr_model* m, int32_t* log_dev, int32_t n_steps_cap, int32_t n_graphs_cap);
/* (ABI 6) The companion read-out: ties[(s * 6 + k) * n_graphs_cap + g] = the number of candidate pairs of graph g in edge set k whose distance lies within
 * `tol` (Angstrom, > 0) of the set's hard cutoff at step s -- the pairs at which two runs that differ by rounding (another DBFR_GEMM mode, another
 * batch of the reference) may build different graphs.  A zero row means the step's graph is decided by margins above tol; tools/validate_checkpoint.py
 * and tests/test_examples.py read it next to the edge counts.  Diagnostic: its kernel is launched only while a log is attached.  Same layout, capacity
 * rule and switch-off (log == NULL) as dbfr_model_set_edge_log.                                                                                */
int dbfr_model_set_tie_log(dbfr_model* m, int32_t* log_dev, int32_t n_steps_cap, int32_t n_graphs_cap, float tol);

/* ---- pose initialisation, on the device.
 * Replaces the per-pose real-time transforms LigInit + SCProtInit +
 * Atom14ToAllAtomsRepr (druglib/datasets/Docking/struct_init.py:16-53,114-138;
 * formatting.py:41-51) for every graph of an assembled batch at once:
 *   lig_pos       <- torsion kicks in bond order, (x - centroid) R^T + tr
 *   torsion_angle <- chi_k = sc_u * sc_torsion_edge_mask (psi kept)
 *   rec_pos       <- side chains rebuilt from the templates, compacted
 * The random draws come from the caller as a device tape in the reference's
 * draw order per pose.  b->lig_pos must hold the ligands' input conformers.     */
typedef struct {
  const float* tor_u;  /* [NTOR]  U(-pi,pi) torsion kicks                       */
  const float* rot;    /* [G,3,3] uniformly random rotation matrices, row-major  */
  const float* tr;     /* [G,3]   N(0, tr_sigma_max) translations                */
  const float* sc_u;   /* [NR,4]  U(-pi,pi) chi draws (unmasked)                 */
} dbfr_init_tape;

int dbfr_init_poses(const dbfr_model* m, const dbfr_batch* b, const dbfr_init_tape* tape, float* atom14_out,
                    void* hip_stream);

/* ---- pocket templates, on the device.
 * Replaces extract_chi_and_template (druglib/utils/obj/prot_math.py:116-241,
 * called once per pocket from SCPocketFinderDefault, pocket_pipeline.py:174-189):
 * from residue types and atom14 coordinates (unused slots zero) to the backbone
 * frames, psi/chi1..4 (radians), the per-residue default frames [n,8,4,4] and the
 * atoms' positions inside their rigid groups [n,14,3] -- the inverse of the
 * side-chain rebuild inside dbfr_sample.  All pointers are device pointers; the
 * residue tables are compiled in.  aatype in [0, 20].                            */
int dbfr_extract_templates(int32_t n_res, const int32_t* aatype, const float* atom14_pos, float* backbone_transl,
                           float* backbone_rots, float* default_frame, float* rigid_group_positions,
                           float* torsion_angle, void* hip_stream);

/* Binding-site residues of n_prot proteins in one pass (ahead of row f2): what
 * Protein.query_region / select_bs computes (druglib/utils/obj/protein.py:154-240,
 * druglib/utils/bio_utils/select_pocket.py:12-99; SCPocketFinderDefault uses mode
 * 'any', cutoff 12, all ligand atoms, pocket_pipeline.py:147-161).  Residue r of
 * protein p (rows res_ptr[p]..res_ptr[p+1]) is selected iff the squared distance
 * between one of its present atoms and one of the protein's reference points
 * (rows ref_ptr[p]..ref_ptr[p+1] of ref_pos) is <= cutoff^2; the nearest residue is
 * always selected; max_neighbors > 0 keeps only that many nearest selected
 * residues.  atoms_per_res = 37 / 14 (mode 'any'), a column subset (atom modes) or
 * 1 (centroids).  All pointers are device pointers; min_dist2 [n_res_total] and
 * res_mask [n_res_total] (0/1 bytes) are outputs.                                   */
int dbfr_select_pocket(int32_t n_prot, int32_t n_res_total, const int32_t* res_ptr, int32_t atoms_per_res,
                       const float* atom_pos, const float* atom_mask, const int32_t* ref_ptr, const float* ref_pos,
                       double cutoff, int32_t max_neighbors, float* min_dist2, uint8_t* res_mask, void* hip_stream);

/* ---- output side: what `complex_modeling`
 * (DiffBindFR/evaluation/export.py:106-312) does with the trajectories dbfr_sample
 * returns -- the per-pose metrics and the PDB text of every pose.                  */

/* Per-(pose, frame) metrics over the trajectories of ONE complex, computed where the
 * trajectories already are (device pointers, fp32):
 *   centroid  |mean(lig + c) - mean(lig_target)|          metrics/centroid.py:6-14
 *   sc_rmsd   side-chain RMSD, best of the two namings of the pi-symmetric groups,
 *             mean over residues with a side chain        metrics/scrmsd.py:64-89
 *   delta_chi |chi_pred - chi_target| per residue and chi (radians, wrapped as the
 *             reference wraps it, best of the pi-periodic alternatives, 0 where the
 *             chi does not exist)                         metrics/angbin.py:11-103
 *   chi_rate  fraction of existing chi_k with delta below chi_bound (15 degrees in
 *             the reference)                              evaluation/export.py:176-181
 *   lig_rmsd  heavy-atom RMSD, minimum over the automorphisms `perms` of the ligand
 *             graph                                       metrics/lrmsd.py:311-335
 * center[3] (host) is the pocket centre the sampler's coordinates are relative to
 * (add_center_pos, common/inference_dataset.py:57-63): it is added to both
 * trajectories and to atom14_target for sc_rmsd; lig_target is absolute.           */
typedef struct {
  int32_t n_pose, n_frame, n_lig, n_res;
  const float*   lig_traj;            /* [n_pose, n_frame, n_lig, 3]                   */
  const float*   prot_traj;           /* [n_pose, n_frame, n_res, 14, 3]               */
  const float*   lig_target;          /* [n_lig, 3]                                    */
  const float*   atom14_target;       /* [n_res, 14, 3] pocket-centred like prot_traj  */
  const float*   atom14_target_mask;  /* [n_res, 14] 0/1                               */
  const int32_t* aatype;              /* [n_res] in [0, 20]                            */
  int32_t        n_perm;              /* automorphisms (>= 1 when lig_rmsd is asked)   */
  const int32_t* perms;               /* [n_perm, n_lig]: atom perms[p][a] of the pose is compared with target atom a */
  const int32_t* heavy_mask;          /* [n_lig] 0/1 (atoms that count), or NULL = all */
  float          center[3];
  float          chi_bound;           /* radians                                       */
} dbfr_pose_metrics_in;

typedef struct {                      /* any pointer may be NULL = not wanted          */
  float* centroid;                    /* [n_pose, n_frame]                             */
  float* sc_rmsd;                     /* [n_pose, n_frame]                             */
  float* chi_rate;                    /* [n_pose, n_frame, 4]                          */
  float* delta_chi;                   /* [n_pose, n_frame, n_res, 4]                   */
  float* lig_rmsd;                    /* [n_pose, n_frame]                             */
} dbfr_pose_metrics_out;

int dbfr_pose_metrics(const dbfr_pose_metrics_in* in, const dbfr_pose_metrics_out* out, void* hip_stream);

/* PDB text of a protein (host code, all pointers host): byte-for-byte what
 * Protein.pos_update(pos14).to_pdb() writes (druglib/utils/obj/protein.py:478-537,
 * 678-800).  The static part of a structure is the topology; a pose replaces the
 * atom14 coordinates of `rows` (the pocket residues inside the protein,
 * evaluation/export.py:261-268).                                                  */
typedef struct {
  int32_t        n_res;
  const int32_t* aatype;              /* [n_res] in [0, 20] (20 = 'UNK')               */
  const float*   atom37_pos;          /* [n_res, 37, 3] coordinates of the input structure */
  const float*   atom37_mask;         /* [n_res, 37] atoms present (>= 0.5)            */
  const int32_t* residue_index;       /* [n_res] PDB residue numbers                   */
  const int32_t* chain_index;         /* [n_res] 0-based chain ids (A..Z, AA, BA, ...) */
  const double*  b_factors;           /* [n_res, 37]                                   */
  const char*    remark;              /* first line, or NULL for none                  */
} dbfr_pdb_topology;

/* Formats one structure into out (capacity cap bytes, no terminating NUL needed) and
 * returns the byte count; if cap is too small nothing is written and the required
 * count is returned.  n_rows == 0: the topology's own coordinates.  rows == NULL
 * with n_rows == n_res: pos14 covers every residue in order.  model < 0: no ENDMDL. */
int64_t dbfr_pdb_format(const dbfr_pdb_topology* topo, int32_t n_rows, const int32_t* rows, const float* pos14,
                        int32_t model, int32_t add_end, char* out, int64_t cap);

/* Writes n_pose files: paths[i] receives the structure with pos14[i] ([n_pose, n_rows,
 * 14, 3]) on n_threads host threads (<= 0: one per pose up to the core count).      */
int dbfr_pdb_write_files(const dbfr_pdb_topology* topo, int32_t n_rows, const int32_t* rows, const float* pos14,
                         int32_t n_pose, const char* const* paths, int32_t n_threads);

/* SD file of a ligand pose (host code, all pointers host): what the reference writes per pose as `lig_final.sdf`
 * (DiffBindFR/evaluation/export.py:97-103,236-244: Ligand3D.pos_update(pose) -> Chem.SDWriter).  The V2000 mol block of
 * the (hydrogen-free) ligand is prepared once as text; per pose only the coordinate columns change.
 * Byte parity with RDKit's SDWriter is NOT pinned (RDKit is absent offline): the block follows the CTfile V2000 layout
 * RDKit reads and writes (%10.4f coordinates), header program line "DBFR-HIP".                                        */
typedef struct {
  int32_t            n_atoms;    /* atoms of the block = rows of a pose                                             */
  const char*        header;     /* 3 header lines + counts line, each newline-terminated                            */
  const char* const* atom_tail;  /* [n_atoms] atom line after the 30 coordinate columns (" C   0  0 ...", no newline) */
  const char*        trailer;    /* bond block, property block, "M  END", data items, "$$$$", newline-terminated      */
} dbfr_sdf_template;

/* Formats one pose (pos [n_atoms,3]) into out (capacity cap) and returns the byte count; too small a cap writes nothing. */
int64_t dbfr_sdf_format(const dbfr_sdf_template* t, const float* pos, char* out, int64_t cap);
/* paths[i] receives pose i of pos [n_pose, n_atoms, 3] on n_threads host threads (<= 0: one per pose up to the cores). */
int dbfr_sdf_write_files(const dbfr_sdf_template* t, const float* pos, int32_t n_pose, const char* const* paths,
                         int32_t n_threads);

/* ---- MDN pose scorer: the network forward of the KarmaDock scorer the reference runs on the
 * sampled poses (DiffBindFR/scoring/architecture/KarmaDock_sc.py:58-101, called from DiffBindFR/common/engines.py:230-302):
 * graph-transformer ligand encoder, GVP pocket encoder, mixture-density head, score = sum over (ligand atom, residue)
 * pairs within 5 A of the 10-component mixture density at their distance.  Inputs are the featurised tensors the
 * reference's HeteroData batch holds (the RDKit / openfold featurisation stays on the host, outside this path).       */
typedef struct dbfr_mdn_model dbfr_mdn_model;
/* tensors: the reference module's state_dict entries (names relative to KarmaDock: "lig_encoder...", "pro_encoder...",
 * "mdn_layer..."); tensors of modules the scoring forward never calls (egnn_layers, gates, ...) may be present.        */
int  dbfr_mdn_model_create(const dbfr_tensor* tensors, int32_t n_tensors, dbfr_mdn_model** out);
void dbfr_mdn_model_destroy(dbfr_mdn_model* m);

typedef struct {                      /* device pointers; graphs are contiguous node ranges                              */
  int32_t B, NL, EL, NR, EP;          /* graphs, ligand atoms, directed covalent ligand edges, residues, pocket edges     */
  const int32_t* lig_ptr;             /* [B+1]                                                                            */
  const float*   lig_node_s;          /* [NL,89]   data['ligand'].node_s                                                  */
  const float*   lig_edge_s;          /* [EL,20]   edge_s[cov_edge_mask]                                                  */
  const int32_t* lig_edge_src;        /* [EL]      edge_index[0] (row)                                                    */
  const int32_t* lig_edge_dst;        /* [EL]      edge_index[1] (col): attention is normalised over the edges into col   */
  const int32_t* lig_in_ptr;          /* [NL+1]    CSR over lig_in_edge by col                                            */
  const int32_t* lig_in_edge;         /* [EL]      edge ids grouped by col, ascending inside a group                      */
  const float*   lig_pos;             /* [NL,3]    data['ligand'].xyz (the pose)                                          */
  const float*   lig_s_in;            /* [NL,128]  optional: ligand embeddings computed before (they do not depend on the
                                                   pose), NULL = run the ligand encoder                                   */
  const int32_t* res_ptr;             /* [B+1]                                                                            */
  const float*   pro_node_s;          /* [NR,9]                                                                           */
  const float*   pro_node_v;          /* [NR,3,3]                                                                         */
  const int32_t* pro_edge_src;        /* [EP]      edge_index[0] (message source j)                                       */
  const int32_t* pro_edge_dst;        /* [EP]      edge_index[1] (target i); edges GROUPED BY TARGET (knn_graph order)    */
  const int32_t* pro_in_ptr;          /* [NR+1]    CSR over the edge array by target                                      */
  const float*   pro_edge_s;          /* [EP,21]                                                                          */
  const float*   pro_edge_v;          /* [EP,1,3]                                                                         */
  const int32_t* pro_seq;             /* [NR]      residue type ids (< 31)                                                */
  const float*   pro_xyz_full;        /* [NR,14,3] atom14 coordinates of the pose (unused slots as the featuriser leaves them) */
  float          dist_threshold;      /* pairs farther apart contribute 0 (KarmaDock.forward passes 5.0); <= 0 = 5.0       */
} dbfr_mdn_batch;

int dbfr_mdn_workspace_bytes(const dbfr_mdn_batch* b, size_t* bytes);
/* score [B]; lig_s_out [NL,128] / pro_s_out [NR,128] optional (embeddings, e.g. to reuse lig_s for the other poses).      */
int dbfr_mdn_forward(dbfr_mdn_model* m, const dbfr_mdn_batch* b, float* score, float* lig_s_out, float* pro_s_out,
                     void* workspace, size_t workspace_bytes, void* hip_stream);

/* The pocket half of the scorer's input on the device, for n_graph pockets / poses at once: what `get_protein_feature`
 * (DiffBindFR/scoring/dataset/protein_feature.py:137-216) computes behind its PDB parser from residue types (< 20) and
 * atom14 coordinates with absent atoms at the origin (= dbfr_sample's atom14 output + the pocket centre): node_s [n_res,9],
 * node_v [n_res,3,3], the topk (<= 32; the reference uses 30) nearest CA neighbours of every residue as edges j -> i grouped
 * by i (edge_src / edge_dst / in_ptr [n_res+1]), edge_s [E,21], edge_v [E,3].  edge_ptr [n_graph+1]: first edge of every
 * pocket, E_g = n_g * min(topk, n_g - 1) (the caller knows the residue counts).  Pockets of at most 1024 residues.        */
int dbfr_mdn_pocket_features(int32_t n_graph, int32_t n_res, const int32_t* res_ptr, const int32_t* edge_ptr,
                             const int32_t* aatype, const float* atom14_pos, int32_t topk, float* node_s, float* node_v,
                             int32_t* edge_src, int32_t* edge_dst, int32_t* in_ptr, float* edge_s, float* edge_v,
                             void* hip_stream);

/* Synchronises the stream and returns the device-side status word of the last
 * dbfr_score / dbfr_sample issued with this workspace (DBFR_OK, DBFR_ERR_CAPACITY,
 * DBFR_ERR_NUMERIC).  counters (may be NULL) receives [8] int64: edges of the last
 * step in the order {lig, atom, cross lig<-atom, center, tor, sc_tor, cross atom<-lig, 0}. */
int dbfr_status_sync(void* workspace, void* hip_stream, int64_t* counters);

/* Which matrix instruction carries the 144 x W GEMM of the radial MLP (97-99 % of the arithmetic) in the K=144 convs.
 * All modes produce fp32 results from fp32 weights and fp32 activations:
 *   DBFR_GEMM_F32            v_mfma_f32_16x16x4_f32 (fp32 operands): k_conv / k_conv2;
 *   (1, DBFR_GEMM_SPLIT_BF16 -- three bf16 pieces per operand, six products on v_mfma_f32_16x16x32_bf16, kernel k_conv2r: round 2's default,
 *                            retired with ABI 5 once no conv needed it as a fall-back; the number stays unused)
 *   DBFR_GEMM_SPLIT_F16      every operand cut into TWO fp16 pieces (hi = fp16(x), lo = fp16(x - hi): 23 of fp32's 24 significand
 *                            bits) after an exact power-of-two scaling that keeps the pieces inside fp16's exponent range (W2: per
 *                            tensor-product run, at model creation; activations: per edge, in the kernel), three partial products
 *                            hi*lo + lo*hi + hi*hi on v_mfma_f32_16x16x32_f16, the two small ones and the large one in separate
 *                            fp32 accumulators.  Kernel k_conv2h (csrc/conv2h.hip);
 *                            accuracy table: profiles/r3_split_experiments.txt.
 *   DBFR_GEMM_REDUCE_FIRST   the arithmetic of SPLIT_F16 with the ORDER of the work changed for the rows of lin.3 that feed a scalar (l = 0)
 *                            output irrep (74 % of the rows at depth 3, all rows of the torsion convs): a scalar message element is linear in
 *                            y (x) h (y = the tensor-product input coupled with the harmonics, h = the hidden layer), and so is the scatter over
 *                            the edges of a target node, so Z[t,c,k] = sum_{e -> t} y[e,c] h[e,k] is formed first and the 144 x W GEMM runs
 *                            once per TARGET SEGMENT (the edges of one target inside a chunk = <= 32 consecutive edges and <= 4 targets of one graph), not once per
 *                            edge: 8-10 x fewer matrix instructions for those rows.  Kernel k_convz (csrc/convz.hip); the l = 1 outputs stay
 *                            per edge on k_conv2h.  Every lin.3 output row carries its own power-of-two factor there (no row-depth limit).
 *                            MESSAGE BUFFER in this mode: the scalar columns of a segment's FIRST message row hold the segment's SUM; those of
 *                            its other rows are zero in the buffer dbfr_test_conv2 fills (the hook clears it first) and NOT WRITTEN inside the
 *                            sampler, whose reductions read them of segment-first rows only (vector columns: per edge as before).  Chunks
 *                            are cut per graph, by the graph's own targets: what is summed with what never depends on batch mates.
 * The initial mode is DBFR_GEMM_DEFAULT unless the environment variable DBFR_GEMM (f32 | split_f16 | reduce_first, or the numbers 0 | 3 | 4) says otherwise;
 * any other value -- the retired `split` / `split_l1` / 1 / 2 among them -- makes dbfr_model_create fail with DBFR_ERR_ARG (until ABI 6 it selected the fp32 instruction silently).
 * A workspace is laid out for the mode it was sized in: set the mode before dbfr_workspace_bytes.                       */
#define DBFR_GEMM_F32 0
/* (1 was DBFR_GEMM_SPLIT_BF16, k_conv2r: retired with ABI 5; 2 was DBFR_GEMM_SPLIT_BF16_L1, k_conv2s: retired with ABI 4; the numbers stay unused) */
#define DBFR_GEMM_SPLIT_F16 3
#define DBFR_GEMM_REDUCE_FIRST 4
#define DBFR_GEMM_DEFAULT DBFR_GEMM_REDUCE_FIRST
int dbfr_model_set_gemm(dbfr_model* model, int32_t mode);
/* DBFR_GEMM_SPLIT_F16 holds a weight row to 22 significant bits while the row's largest |w| is within 2^17 of the largest |w| that shares its
 * power-of-two factor -- one per tensor-product run; lin.0: one per matrix.  dbfr_model_create measures every run of every conv; a conv with
 * a deeper row (a trained checkpoint may hold one; seeded weights do not) is packed with one factor per ROW instead, which the kernel takes
 * off the accumulator rows (ABI 5; four more vector instructions per edge block and tile): dbfr_model_rowscaled_convs counts those convs and
 * writes "name:depth;" for each (depth = log2 of the spread that was found).  dbfr_model_fallback_convs counts the convs that even so cannot
 * be held by two fp16 pieces and are served by the fp32-instruction kernel (k_conv2) together with the other convs of their launch (an interaction
 * layer / the two torsion heads): none since ABI 5, unless a bias dwarfs its row by more than 2^48.  names (may be NULL) receives the
 * state_dict prefixes, ';'-terminated each ("atom_conv_layers.3;").                                                                  */
int dbfr_model_rowscaled_convs(const dbfr_model* model, char* names, size_t names_cap);
int dbfr_model_fallback_convs(const dbfr_model* model, char* names, size_t names_cap);
int dbfr_model_get_gemm(const dbfr_model* model);

/* ---- introspection / test hooks */
int         dbfr_abi_version(void);
/* First 16 hex digits of the sha256 over the library's source files at build time (diffbindfr_amd/build.py:
 * source_hash): lets a caller check that a prebuilt libdbfr.so belongs to the source tree next to it.               */
const char* dbfr_build_id(void);
const char* dbfr_last_error(void);
/* Real-basis Wigner-3j tensor the library derives (Racah formula) to self-check
 * the closed forms baked into the kernels; out has (2l1+1)(2l2+1)(2l3+1) doubles. */
int dbfr_wigner3j(int32_t l1, int32_t l2, int32_t l3, double* out);
/* Path table of a conv: n_paths rows of {i1,i2,io,l1,l2,lo,mul1,mulo,w_off} int32
 * + coeff as float bits in column 9. kind: 0..3 layer convs by depth, 4 final_conv,
 * 5 tor convs.  Returns number of paths (<= max_paths) or a negative status.       */
int dbfr_conv_paths(int32_t kind, int32_t* table10, int32_t max_paths, int32_t* weight_numel);
/* Test hook (host code, no GPU): the tile packer of DBFR_GEMM_SPLIT_F16.  frag = n_tiles x [9 k-steps of 16][64 lanes][4] fp32 MFMA
 * fragments (K = 144), bias = n_tiles x 16; out = n_tiles x 9280 bytes: [hi, lo][4 k-steps of 32][64 lanes][8 fp16], then
 * [64 lanes][hi 4 | lo 4 fp16] for the last 16 k, then the 16 bias values (fp32) -- everything multiplied by 2^k, the power of two that
 * puts the largest |value| of these tiles into [2^14, 2^15).  Returns k in *k_out.                                                     */
/* Test hook (GPU): the chunk table k_convz walks, for one flat target-sorted edge list cut every `span` edges as if those were graphs: a chunk =
 * consecutive edges, at most 32 of them and at most four targets.  All pointers are device pointers; scratch holds (max_edges + span - 1) / span + 1
 * ints and receives the first chunk of every span, the total behind them; chunk_es[ch] = first edge, chunk_gl[ch] = number of edges (graph 0). */
int dbfr_test_chunk_table(const int32_t* tgt, const int32_t* n_edges_dev, int32_t max_edges, int32_t span, int32_t* scratch, int32_t cap,
                          int32_t* chunk_es, int32_t* chunk_gl, void* hip_stream);
int dbfr_test_pack_f16_tiles(const float* frag, const float* bias, int32_t n_tiles, void* out, int32_t* k_out);
/* The same packer with one power of two per ROW on top of 2^k (ABI 5; what a conv with rows further apart than 2^17 is packed with):
 * rinv_out [n_tiles x 16] receives 2^-d(row), the factor the kernel takes off the accumulator row; *depth_out the row depth that is left. */
int dbfr_test_pack_f16_rows(const float* frag, const float* bias, int32_t n_tiles, void* out, int32_t* k_out, float* rinv_out, int32_t* depth_out);
/* The range guard of the same packer (host code, no GPU): the largest ROW DEPTH d of these tiles taken as one run -- the row's largest
 * |value| is 2^(15 - d) after the run's factor -- and the limit up to which two fp16 pieces hold a row to 22 significant bits (17): a conv
 * with a deeper row in one of its runs is what dbfr_model_fallback_convs reports.                                                     */
int dbfr_test_pack_f16_depth(const float* frag, const float* bias, int32_t n_tiles, int32_t* depth_out, int32_t* depth_ok_out);
/* Profiling hooks: time (ms) spent in the dominant fused conv kernel and the
 * number of launches + edges since the last reset, measured with hip events on
 * the launch stream when profiling is enabled.  conv_flops = algorithmic FLOP
 * (2K(K+W) per edge); ref_form_bytes = HBM bytes the reference's two-kernel form
 * of the same launches would move (4(W+D_in+9)+16 per edge).        */
/* on = 1: events + counters, the independent convs of a layer serialised on the caller's stream so that each is timed
 * alone; on = 2: counters only (launch pattern as in production); 0: off.                                        */
int dbfr_profile_enable(dbfr_model* m, int32_t on);
int dbfr_profile_read(dbfr_model* m, double* conv_ms, int64_t* conv_launches, double* conv_flops,
                      double* ref_form_bytes, int32_t reset);
/* HBM bytes the FUSED conv has to move for the launches the last dbfr_profile_read reported (read before its reset):
 * 4 (48 + 9 + 3 + 48 + 48 + D_in + D_out) per edge = edge record, two gathered radial-MLP rows, gathered input row, message. */
int dbfr_profile_fused_bytes(const dbfr_model* m, double* fused_form_bytes);
/* Flops the matrix pipe EXECUTED in the launches the last dbfr_profile_read reported: per edge `products` x 2 x 144 x (144 + rows walked) in the per-edge
 * kernels (products: 1 fp32 instruction, 3 two-piece fp16, 6 three-piece bf16), the instructions k_convz issued x 16384 in DBFR_GEMM_REDUCE_FIRST.   */
int dbfr_profile_executed_flops(const dbfr_model* m, double* executed_flops);
/* (ABI 6) Of those, the flops that are NOT padding -- *useful_flops: in DBFR_GEMM_REDUCE_FIRST the hidden layer once per edge (the pair of kernels computes it
 * twice), step A's products over the edges a segment holds (k_convz issues them over all 32 slots of a chunk for each of four segment slots, filled or
 * not), step B's over the segments a unit holds (not the 16-column blocks) and the (path, u) pairs that exist (not the padding of a c tile), times the three
 * partial products; in the per-edge modes the executed flops.  *form_bytes: HBM bytes the form that RUNS has to move for the same launches (the fused
 * form's of dbfr_profile_fused_bytes, but in DBFR_GEMM_REDUCE_FIRST the edge records and gathered rows once per kernel of the pair and the
 * scalar-output message columns once per segment).  Either pointer may be null.                                                          */
int dbfr_profile_useful_flops(const dbfr_model* m, double* useful_flops, double* form_bytes);

/* What the fp16 matrix pipe of the CURRENT device sustains: a bare stream of v_mfma_f32_16x16x32_f16 (the instruction of DBFR_GEMM_SPLIT_F16)
 * with random operands on every compute unit, two waves per SIMD, for `seconds` (0 < seconds <= 60; the rate is taken over the second half,
 * when the firmware has settled the clock at the board's power cap).  *tflops = executed TFLOP/s.  bench.py's roofline.frac_of_sustained
 * divides by it; the instruction's nominal peak (2 500) is reached with all-zero operands only.  Blocks the calling thread.            */
int dbfr_probe_mfma_f16(double seconds, double* tflops, void* hip_stream);

/* Test hook: names (';'-separated) / byte offsets / sizes of the library's internal
 * buffers inside the workspace for this batch shape.  Returns the entry count.       */
int dbfr_workspace_layout(const dbfr_model* m, const dbfr_batch* b, const dbfr_limits* lim, char* names,
                          size_t names_cap, size_t* offsets, size_t* bytes, int32_t max_entries);

/* Unit-test hooks: run ONE fused tensor-product conv (k_conv) / ONE segmented-mean +
 * LayerNorm (k_reduce_ln) of the packed model on caller-supplied device buffers.
 * layer >= 0: interaction layer, family 0 lig, 1 cross_al, 2 atom, 3 cross_la;
 * layer -1 final_conv, -2 tor_bond_conv, -3 sc_tor_bond_conv.                          */
int dbfr_test_conv(dbfr_model* m, int32_t layer, int32_t family, int32_t n_edges, const int32_t* n_edges_dev,
                   const int32_t* tgt, const int32_t* gth, const float* emb, const float* sh, const float* tab1,
                   int32_t ld1, const int32_t* idx1, const float* tab2, int32_t ld2, const int32_t* idx2,
                   const float* x, int32_t ldx, float* msg, void* hip_stream);
/* The same conv through the second-generation kernel k_conv2 (persistent, one W2 stream per CU; K=144 convs only).    */
int dbfr_test_conv2(dbfr_model* m, int32_t layer, int32_t family, int32_t n_edges, const int32_t* n_edges_dev,
                    const int32_t* tgt, const int32_t* gth, const float* emb, const float* sh, const float* tab1,
                    int32_t ld1, const int32_t* idx1, const float* tab2, int32_t ld2, const int32_t* idx2,
                    const float* x, int32_t ldx, float* msg, void* hip_stream);
int dbfr_test_reduce_ln(dbfr_model* m, int32_t layer, int32_t family, const float* msg, const int32_t* row_start,
                        const int32_t* row_cnt, int32_t n_nodes, const float* old, int32_t d_old, float* out,
                        int32_t mode, void* hip_stream);
/* (ABI 6) The same reduction with the message interface of DBFR_GEMM_REDUCE_FIRST: seg_first [n_edges] bytes, 1 = the row's scalar-output columns hold a
 * segment's sum (k_convz stores it in the segment's first row and writes nothing into those columns of the other rows: they are not read, whatever
 * they hold); vector-output columns are per edge.  seg_first == NULL: as dbfr_test_reduce_ln.  K = 144 convs only.                              */
int dbfr_test_reduce_ln2(dbfr_model* m, int32_t layer, int32_t family, const float* msg, const int32_t* row_start,
                         const int32_t* row_cnt, int32_t n_nodes, const float* old, int32_t d_old, float* out,
                         int32_t mode, const uint8_t* seg_first, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* DBFR_H */
