"""ctypes binding of libdbfr.so (include/dbfr.h).  Fails loudly when the HIP
library is missing: there is no CPU fallback in the product path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DBFR_LIB") or os.path.join(HERE, "libdbfr.so")      # DBFR_LIB: developer override (kernel variants)

DBFR_OK = 0
DBFR_ERR_CAPACITY = -3
ERRORS = {-1: "DBFR_ERR_ARG", -2: "DBFR_ERR_HIP", -3: "DBFR_ERR_CAPACITY", -4: "DBFR_ERR_SELFTEST",
          -5: "DBFR_ERR_NUMERIC"}

i32, f32, vp = C.c_int32, C.c_float, C.c_void_p


class ModelCfg(C.Structure):
    _fields_ = [(n, i32) for n in ("ns", "nv", "sh_lmax", "num_conv_layers", "lig_node_features",
                                   "lig_edge_features", "distance_embed_dim", "sigma_embed_dim")] + \
               [(n, f32) for n in ("emb_scale", "lig_cutoff", "atom_cutoff", "cross_cutoff", "center_max_distance")] + \
               [(n, i32) for n in ("atom_max_neighbors", "lig_max_neighbors", "dynamic_max_cross", "scale_by_sigma",
                                   "no_sc_torsion")]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", vp), ("numel", C.c_int64)]


_BATCH_INTS = ("G", "NL", "NA", "NR", "EB", "NTOR", "NSC", "max_nl", "max_na", "max_nr")
_BATCH_PTRS = ("lig_ptr", "lig_node", "lig_pos", "bond_src", "bond_dst", "bond_feat", "bond_ptr", "tor_ptr",
               "tor_bond", "rot_mask", "rot_mask_off", "atm_ptr", "res_ptr", "pocket_feat", "rec_pos", "sequence",
               "backbone_transl", "backbone_rots", "default_frame", "rigid_group_positions", "torsion_angle",
               "atom14_slot", "sc_res_chi", "sc_bond", "sc_ptr")


class Batch(C.Structure):
    _fields_ = [(n, i32) for n in _BATCH_INTS] + [(n, vp) for n in _BATCH_PTRS]


class Limits(C.Structure):
    _fields_ = [("aa_avg_neighbors", i32), ("cross_avg_neighbors", i32)]


class Cond(C.Structure):
    _fields_ = [(n, vp) for n in ("t", "tr_sigma", "rot_score_norm", "tor_score_norm2", "sc_tor_score_norm2")]


class Scores(C.Structure):
    _fields_ = [(n, vp) for n in ("tr", "rot", "tor", "sc_tor")]


class Step(C.Structure):
    _fields_ = [(n, f32) for n in ("t", "dt", "tr_sigma", "rot_score_norm", "tor_score_norm2", "tr_g2", "tr_gsdt",
                                   "rot_g2", "rot_gsdt", "tor_g2", "tor_gsdt", "sc_g2", "sc_gsdt")]


class Noise(C.Structure):
    _fields_ = [(n, vp) for n in ("z_tr", "z_rot", "z_tor", "z_sc")]


class InitTape(C.Structure):
    _fields_ = [(n, vp) for n in ("tor_u", "rot", "tr", "sc_u")]


class PoseMetricsIn(C.Structure):
    _fields_ = [("n_pose", C.c_int32), ("n_frame", C.c_int32), ("n_lig", C.c_int32), ("n_res", C.c_int32),
                ("lig_traj", C.c_void_p), ("prot_traj", C.c_void_p), ("lig_target", C.c_void_p),
                ("atom14_target", C.c_void_p), ("atom14_target_mask", C.c_void_p), ("aatype", C.c_void_p),
                ("n_perm", C.c_int32), ("perms", C.c_void_p), ("heavy_mask", C.c_void_p),
                ("center", C.c_float * 3), ("chi_bound", C.c_float)]


class PoseMetricsOut(C.Structure):
    _fields_ = [("centroid", C.c_void_p), ("sc_rmsd", C.c_void_p), ("chi_rate", C.c_void_p), ("delta_chi", C.c_void_p),
                ("lig_rmsd", C.c_void_p)]


_MDN_PTRS = ("lig_ptr", "lig_node_s", "lig_edge_s", "lig_edge_src", "lig_edge_dst", "lig_in_ptr", "lig_in_edge", "lig_pos",
             "lig_s_in", "res_ptr", "pro_node_s", "pro_node_v", "pro_edge_src", "pro_edge_dst", "pro_in_ptr", "pro_edge_s",
             "pro_edge_v", "pro_seq", "pro_xyz_full")


class MdnBatch(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "NL", "EL", "NR", "EP")] + [(n, C.c_void_p) for n in _MDN_PTRS] + \
               [("dist_threshold", C.c_float)]


class SdfTemplate(C.Structure):
    _fields_ = [("n_atoms", C.c_int32), ("header", C.c_char_p), ("atom_tail", C.POINTER(C.c_char_p)), ("trailer", C.c_char_p)]


class PdbTopology(C.Structure):
    _fields_ = [("n_res", C.c_int32), ("aatype", C.c_void_p), ("atom37_pos", C.c_void_p), ("atom37_mask", C.c_void_p),
                ("residue_index", C.c_void_p), ("chain_index", C.c_void_p), ("b_factors", C.c_void_p),
                ("remark", C.c_char_p)]


# every symbol include/dbfr.h declares (tests check that the library exports all of them)
SYMBOLS = ["dbfr_model_create", "dbfr_model_destroy", "dbfr_model_set_edge_log", "dbfr_model_set_tie_log", "dbfr_model_fallback_convs", "dbfr_model_rowscaled_convs", "dbfr_model_set_gemm", "dbfr_model_get_gemm", "dbfr_workspace_bytes", "dbfr_score", "dbfr_sample",
           "dbfr_sample_range", "dbfr_capacity_report",
           "dbfr_init_poses", "dbfr_extract_templates", "dbfr_status_sync", "dbfr_abi_version", "dbfr_build_id", "dbfr_last_error", "dbfr_wigner3j", "dbfr_conv_paths", "dbfr_test_pack_f16_tiles", "dbfr_test_pack_f16_rows", "dbfr_test_chunk_table", "dbfr_test_pack_f16_depth", "dbfr_probe_mfma_f16",
           "dbfr_profile_enable", "dbfr_profile_read", "dbfr_profile_fused_bytes", "dbfr_profile_executed_flops", "dbfr_profile_useful_flops", "dbfr_workspace_layout", "dbfr_test_conv", "dbfr_test_conv2", "dbfr_test_reduce_ln", "dbfr_test_reduce_ln2",
           "dbfr_pose_metrics", "dbfr_pdb_format", "dbfr_pdb_write_files", "dbfr_select_pocket", "dbfr_sdf_format",
           "dbfr_sdf_write_files", "dbfr_mdn_model_create", "dbfr_mdn_model_destroy", "dbfr_mdn_workspace_bytes", "dbfr_mdn_forward", "dbfr_mdn_pocket_features"]

_lib = None


class DbfrError(RuntimeError):
    pass


def load():
    """Load libdbfr.so; raises if it has not been built (python -m diffbindfr_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DbfrError(f"{LIB_PATH} is missing: build the HIP library first "
                        f"(python -m diffbindfr_amd.build). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.dbfr_last_error.restype = C.c_char_p
    lib.dbfr_build_id.restype = C.c_char_p
    lib.dbfr_model_create.argtypes = [C.POINTER(ModelCfg), C.POINTER(Tensor), i32, C.POINTER(vp)]
    lib.dbfr_model_destroy.argtypes = [vp]
    lib.dbfr_model_destroy.restype = None
    lib.dbfr_model_set_edge_log.argtypes = [vp, vp, i32, i32]
    lib.dbfr_model_set_tie_log.argtypes = [vp, vp, i32, i32, C.c_float]
    lib.dbfr_model_fallback_convs.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.dbfr_model_rowscaled_convs.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.dbfr_model_set_gemm.argtypes = [vp, i32]
    lib.dbfr_model_get_gemm.argtypes = [vp]
    lib.dbfr_workspace_bytes.argtypes = [vp, C.POINTER(Batch), C.POINTER(Limits), C.POINTER(C.c_size_t)]
    lib.dbfr_score.argtypes = [vp, C.POINTER(Batch), C.POINTER(Cond), C.POINTER(Scores), vp, C.c_size_t,
                               C.POINTER(Limits), vp]
    lib.dbfr_sample.argtypes = [vp, C.POINTER(Batch), C.POINTER(Step), i32, C.POINTER(Noise), vp, vp, vp, vp,
                                C.c_size_t, C.POINTER(Limits), vp]
    lib.dbfr_sample_range.argtypes = [vp, C.POINTER(Batch), C.POINTER(Step), i32, i32, C.POINTER(Noise), vp, vp, vp, vp,
                                      C.c_size_t, C.POINTER(Limits), vp]
    lib.dbfr_capacity_report.argtypes = [vp, vp, C.POINTER(i32), C.POINTER(C.c_int64)]
    lib.dbfr_init_poses.argtypes = [vp, C.POINTER(Batch), C.POINTER(InitTape), vp, vp]
    lib.dbfr_extract_templates.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dbfr_status_sync.argtypes = [vp, vp, C.POINTER(C.c_int64)]
    lib.dbfr_wigner3j.argtypes = [i32, i32, i32, C.POINTER(C.c_double)]
    lib.dbfr_conv_paths.argtypes = [i32, C.POINTER(i32), i32, C.POINTER(i32)]
    lib.dbfr_test_pack_f16_tiles.argtypes = [vp, vp, i32, vp, C.POINTER(i32)]
    lib.dbfr_test_pack_f16_depth.argtypes = [vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.dbfr_test_pack_f16_rows.argtypes = [vp, vp, i32, vp, C.POINTER(i32), vp, C.POINTER(i32)]
    lib.dbfr_test_chunk_table.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp, vp]
    lib.dbfr_probe_mfma_f16.argtypes = [C.c_double, C.POINTER(C.c_double), vp]
    lib.dbfr_profile_enable.argtypes = [vp, i32]
    lib.dbfr_profile_fused_bytes.argtypes = [vp, C.POINTER(C.c_double)]
    lib.dbfr_profile_executed_flops.argtypes = [vp, C.POINTER(C.c_double)]
    lib.dbfr_profile_useful_flops.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.dbfr_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), i32]
    lib.dbfr_workspace_layout.argtypes = [vp, C.POINTER(Batch), C.POINTER(Limits), C.c_char_p, C.c_size_t,
                                          C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), i32]
    lib.dbfr_test_conv.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp]
    lib.dbfr_test_conv2.argtypes = lib.dbfr_test_conv.argtypes
    lib.dbfr_test_reduce_ln.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp, i32, vp, i32, vp]
    lib.dbfr_test_reduce_ln2.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp, i32, vp, i32, vp, vp]
    lib.dbfr_select_pocket.argtypes = [i32, i32, vp, i32, vp, vp, vp, vp, C.c_double, i32, vp, vp, vp]
    lib.dbfr_pose_metrics.argtypes = [C.POINTER(PoseMetricsIn), C.POINTER(PoseMetricsOut), vp]
    lib.dbfr_pdb_format.argtypes = [C.POINTER(PdbTopology), i32, vp, vp, i32, i32, vp, C.c_int64]
    lib.dbfr_pdb_format.restype = C.c_int64
    lib.dbfr_pdb_write_files.argtypes = [C.POINTER(PdbTopology), i32, vp, vp, i32, C.POINTER(C.c_char_p), i32]
    lib.dbfr_sdf_format.argtypes = [C.POINTER(SdfTemplate), vp, vp, C.c_int64]
    lib.dbfr_sdf_format.restype = C.c_int64
    lib.dbfr_sdf_write_files.argtypes = [C.POINTER(SdfTemplate), vp, i32, C.POINTER(C.c_char_p), i32]
    lib.dbfr_mdn_model_create.argtypes = [C.POINTER(Tensor), i32, C.POINTER(vp)]
    lib.dbfr_mdn_model_destroy.argtypes = [vp]
    lib.dbfr_mdn_model_destroy.restype = None
    lib.dbfr_mdn_workspace_bytes.argtypes = [C.POINTER(MdnBatch), C.POINTER(C.c_size_t)]
    lib.dbfr_mdn_pocket_features.argtypes = [i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dbfr_mdn_forward.argtypes = [vp, C.POINTER(MdnBatch), vp, vp, vp, vp, C.c_size_t, vp]
    if lib.dbfr_abi_version() != 6:
        raise DbfrError("libdbfr ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != DBFR_OK:
        msg = load().dbfr_last_error().decode()
        raise DbfrError(f"{ERRORS.get(rc, rc)}: {msg}")


def workspace_views(model_handle, batch_c, limits, ws):
    """dict name -> uint8 view of the internal buffer inside the torch workspace (tests/debug)."""
    lib = load()
    names = C.create_string_buffer(8192)
    offs = (C.c_size_t * 128)()
    nbytes = (C.c_size_t * 128)()
    n = lib.dbfr_workspace_layout(model_handle, C.byref(batch_c), C.byref(limits), names, 8192, offs, nbytes, 128)
    if n < 0:
        check(n)
    out = {}
    for i, nm in enumerate(names.value.decode().split(";")[:n]):
        out[nm] = ws[offs[i]:offs[i] + nbytes[i]]
    return out
