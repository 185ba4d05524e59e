"""Host-side logic (no GPU): schedule, packing, registry, state_dict contract, library exports."""
import os
import re

import numpy as np
import pytest
import torch

import diffbindfr_amd as dba
from diffbindfr_amd import dist as ddist, lib as L, schedule as psched, synthetic
from diffbindfr_amd.packing import PackedBatch
from tests.helpers import GOLDEN

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dbfr.h")).read()
    declared = set(re.findall(r"\b(dbfr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = L.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(L.SYMBOLS) == declared
    assert lib.dbfr_abi_version() == 6


def test_gemm_mode_entry_points_reject_bad_arguments():
    """dbfr_model_set_gemm / dbfr_model_get_gemm (include/dbfr.h): argument checks need no GPU."""
    lib = L.load()
    assert lib.dbfr_model_set_gemm(None, 1) == -1 and b"dbfr_model_set_gemm" in lib.dbfr_last_error()
    assert lib.dbfr_model_get_gemm(None) == -1
    hdr = open(os.path.join(ROOT, "include", "dbfr.h")).read()
    modes = dict(re.findall(r"#define (DBFR_GEMM_[A-Z0-9_]+) (\w+)", hdr))
    # (modes 1 and 2, the three-bf16-piece kernels of round 2, are retired: their numbers stay unused)
    assert modes["DBFR_GEMM_F32"] == "0" and "DBFR_GEMM_SPLIT_BF16" not in modes and "DBFR_GEMM_SPLIT_BF16_L1" not in modes
    assert modes["DBFR_GEMM_SPLIT_F16"] == "3" and modes["DBFR_GEMM_REDUCE_FIRST"] == "4"
    assert modes["DBFR_GEMM_DEFAULT"] in ("DBFR_GEMM_SPLIT_F16", "DBFR_GEMM_REDUCE_FIRST")
    from diffbindfr_amd.score_model import GEMM_MODES
    assert sorted(GEMM_MODES.values()) == [0, 3, 4]


def test_product_schedule_matches_reference_fixture():
    z = np.load(os.path.join(GOLDEN, "schedule.npz"))
    recs, arr = psched.steps(psched.sample_cfg())
    assert len(recs) == 20
    for r, row in zip(recs, z["steps"]):
        assert [r.t, r.dt, r.tr_sigma, r.rot_sigma, r.tor_sigma, r.sc_tor_sigma] == list(row[:6])
        assert abs(r.tr_g2 - np.float32(row[6]) ** 2) <= 1e-6 * r.tr_g2
    for i, v in zip(z["so3_idx"], z["so3_val"]):
        assert abs(psched.so3_exp_score_norm(int(i)) - v) <= 1e-9 * max(1.0, abs(v))
    for i, v in zip(z["torus_idx"], z["torus_norm_seed0"]):
        got = psched.torus_score_norm_entry(int(i), 0)
        assert got == v or (np.isnan(got) and np.isnan(v))
    assert recs[-1].noise_free and not recs[0].noise_free
    assert arr[3].dt == np.float32(recs[3].dt)


def test_product_schedule_equals_oracle_schedule():
    from oracle import schedule as osched
    ocfg = osched.default_sample_cfg()
    recs, _ = psched.steps(psched.sample_cfg())
    for i in (0, 7, 19):
        o = osched.step_scalars(ocfg, i)
        r = recs[i]
        assert r.rot_score_norm == float(o.rot_score_norm) and r.tor_score_norm2 == float(o.tor_score_norm2)
        assert r.tr_gsdt == float(o.tr_g * np.sqrt(o.dt)) and r.sc_g2 == float(o.sc_tor_g ** 2)


def test_product_schedule_ode_and_guards():
    """`sample_cfg.type='ode'` (scFlex.py:162-165,199-200): drift 0.5 g^2 (exactly half the SDE's factor), no noise factor, no step draws;
    `no_random` (:167-183): SDE factors, no step draws; `time_schedule` other than 'linear' raises the reference's error (:91)."""
    sde, _ = psched.steps(psched.sample_cfg())
    ode, arr = psched.steps(psched.sample_cfg(dict(type="ode")))
    nor, _ = psched.steps(psched.sample_cfg(dict(no_random=True)))
    assert len(ode) == len(sde) == 20
    for a, b, c in zip(sde, ode, nor):
        for k in ("tr", "rot", "tor", "sc"):
            assert getattr(b, f"{k}_g2") == float(np.float32(0.5) * np.float32(getattr(a, f"{k}_g2"))) and getattr(b, f"{k}_gsdt") == 0.0
            assert getattr(c, f"{k}_g2") == getattr(a, f"{k}_g2") and getattr(c, f"{k}_gsdt") == getattr(a, f"{k}_gsdt")
        assert (b.t, b.dt, b.rot_score_norm, b.tor_score_norm2) == (a.t, a.dt, a.rot_score_norm, a.tor_score_norm2)
        assert b.noise_free and c.noise_free
    assert arr[0].tr_gsdt == 0.0 and arr[5].tr_g2 == np.float32(ode[5].tr_g2)
    zm = np.load(os.path.join(GOLDEN, "sampler_modes.npz"))
    with pytest.raises(NotImplementedError) as e:
        psched.steps(psched.sample_cfg(dict(time_schedule="cosine")))
    assert str(e.value) == str(zm["time_schedule_error"])
    # the drop-in reads the mode from test_cfg.sample_cfg like the reference (scFlex.py:131)
    import diffbindfr_amd as dba
    samp = dba.DiffBindFRHIP(diffusion_model=None, test_cfg=dict(sample_cfg=dict(type="ode")))
    recs, _ = samp.schedule()
    assert all(r.noise_free and r.tr_gsdt == 0.0 for r in recs)
    from diffbindfr_amd.sampler import draw_noise_tape
    g = torch.Generator().manual_seed(3)
    st = g.get_state()
    z = draw_noise_tape(recs, 2, 3, 4, generator=g)
    assert torch.equal(g.get_state(), st) and all(float(v.abs().max()) == 0.0 for v in z.values())      # nothing drawn


def test_state_dict_contract():
    from oracle import score_model as sm
    shapes = sm.param_shapes(sm.default_cfg())
    model = dba.TensorProductModelHIP({})
    sd = model.state_dict()
    assert sorted(sd) == sorted(shapes)                     # SURVEY.md Appendix B.3 key set
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    # e3nn buffers of a reference checkpoint are ignored and reported, anything else is rejected
    extra = dict(sd)
    extra["lig_conv_layers.0.tp.output_mask"] = torch.ones(3)
    extra["final_tp_tor.weight"] = torch.zeros(0)
    model.load_state_dict(extra, strict=True)
    assert sorted(model.ignored_keys) == ["final_tp_tor.weight", "lig_conv_layers.0.tp.output_mask"]
    bad = dict(sd)
    bad["not_a_key"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        model.load_state_dict(bad, strict=True)


def _walk_load(module, state_dict, strict=True):
    """The module-by-module walk of the reference's loader (druglib/core/runner/checkpoint.py:62-93: recursive
    ``_load_from_state_dict`` with strict=True, then the unexpected / missing key check) -- NOT torch's
    ``load_state_dict``, which predict.py never calls."""
    unexpected, missing, errs = [], [], []
    state_dict = state_dict.copy()

    def load(m, prefix=""):
        m._load_from_state_dict(state_dict, prefix, {}, True, missing, unexpected, errs)
        for name, child in m._modules.items():
            if child is not None:
                load(child, prefix + name + ".")
    load(module)
    missing = [k for k in missing if "num_batches_tracked" not in k]
    if strict and (unexpected or missing or errs):
        raise RuntimeError(f"unexpected {unexpected} missing {missing} {errs}")
    return unexpected, missing


def test_reference_loader_fixture():
    """tests/golden/boundary.npz = key list / shapes of a checkpoint that the reference's OWN `load_checkpoint(strict=True,
    drop_keys=['^ema_'])` accepted for `DiffBindFRHIP` built by the reference's own `build_task_model` (frozen by
    tests/golden/make_golden.py::golden_boundary in the build container): e3nn buffers under every `tp` /
    `final_tp_tor`, `ema_` duplicates, DDP's `module.` prefix.  Replayed here on the same walk."""
    import os
    import re
    from collections import OrderedDict
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "boundary.npz"))
    g = torch.Generator().manual_seed(0)
    ck = OrderedDict()
    for k, shp in zip(z["keys"].tolist(), z["shapes"].tolist()):
        shape = tuple(int(x) for x in shp.split(",")) if shp else ()
        ck[k] = torch.rand(shape, generator=g)
    # predict.py:118-125 -> checkpoint.py:442-453: drop ^ema_, strip ^module.
    sd = OrderedDict((re.sub(r"^module\.", "", k), v) for k, v in ck.items() if not re.match(r"^ema_", k))
    model = dba.MLDOCK_BUILDER.build(dict(type="DiffBindFRHIP", diffusion_model=dict(type="TensorProductModelHIP", cfg={})),
                                     default_args=dict(train_cfg=None, test_cfg=None))
    fp0 = model.diffusion_model._fingerprint()
    _walk_load(model, sd, strict=True)
    assert sorted(model.diffusion_model.ignored_keys) == sorted(z["absorbed"].tolist())
    own = model.state_dict()
    assert len(own) == int(z["n_params"])
    for k, v in own.items():
        assert torch.equal(v, sd[k]), k
    assert model.diffusion_model._fingerprint() != fp0           # the packed device model is rebuilt on the next call
    # torch's own loader takes the same checkpoint too
    model.load_state_dict(sd, strict=True)
    # a genuinely unknown key / a missing tensor still fail under strict
    bad = OrderedDict(sd)
    bad["diffusion_model.lig_conv_layers.0.fc.lin.9.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        _walk_load(model, bad, strict=True)
    bad = OrderedDict(sd)
    del bad["diffusion_model.final_conv.fc.lin.0.bias"]
    with pytest.raises(RuntimeError):
        _walk_load(model, bad, strict=True)


def test_registry_keyword_is_overwrite_like_the_reference():
    """druglib/utils/registry.py:285-358: `register_module(name=None, overwrite=False, module=None)`."""
    with pytest.raises(KeyError):
        dba.INTERACTION.register_module(name="TensorProductModelHIP", module=dba.TensorProductModelHIP)
    dba.INTERACTION.register_module(name="TensorProductModelHIP", overwrite=True, module=dba.TensorProductModelHIP)
    with pytest.raises(KeyError):
        dba.INTERACTION.register_module(overwrite="yes")
    with pytest.raises(TypeError):
        dba.INTERACTION.register_module(force=True)


def test_registry_builds_like_the_reference():
    cfg = dict(type="DiffBindFRHIP", diffusion_model=dict(type="TensorProductModelHIP", cfg=dict(ns=48, nv=12)),
               test_cfg=dict(sample_cfg=dict(inference_steps=22, actual_steps=20)))
    m = dba.MLDOCK_BUILDER.build(cfg, default_args=dict(train_cfg=None))
    assert isinstance(m, dba.DiffBindFRHIP) and isinstance(m.diffusion_model, dba.TensorProductModelHIP)
    assert "TensorProductModelHIP" in dba.INTERACTION and dba.INTERACTION.get("nope") is None
    with pytest.raises(KeyError):
        dba.INTERACTION.build(dict(type="nope"))
    # the reference's own type name is redirected to the HIP class (drop-in without editing the config)
    m2 = dba.DiffBindFRHIP(diffusion_model=dict(type="TensorProductModel", cfg={}))
    assert isinstance(m2.diffusion_model, dba.TensorProductModelHIP)


def test_no_cpu_fallback():
    d = synthetic.make_batch(2, n_complex=1, poses=1, seed=0, n_atoms=40, n_lig=8)
    model = dba.TensorProductModelHIP({})
    d.t = torch.ones(1); d.tr_sigma = torch.ones(1); d.rot_score_norm = torch.ones(1, 1)
    d.tor_score_norm2 = torch.ones(int(d.tor_edge_mask.sum())); d.sc_tor_score_norm2 = torch.ones(d.sc_torsion_edge_mask.shape)
    with pytest.raises(L.DbfrError):
        model(d)                                             # CPU tensors: must raise, never compute


def test_packing_invariants():
    d = synthetic.make_batch(2, n_complex=3, poses=2, seed=5, n_atoms=50, n_lig=9)
    pb = PackedBatch(d, "cpu")
    t, dm = pb.t, pb.dims
    assert dm["G"] == 6 and t["lig_ptr"][-1] == dm["NL"] and t["atm_ptr"][-1] == dm["NA"] and t["res_ptr"][-1] == dm["NR"]
    assert torch.all(t["bond_src"][1:] >= t["bond_src"][:-1])                       # CSR by source
    assert t["bond_ptr"][-1] == dm["EB"] and t["tor_ptr"][-1] == dm["NTOR"] and t["sc_ptr"][-1] == dm["NSC"]
    # every torsion row: bond (u,v) with u outside and v inside the rotating side
    for k in range(dm["NTOR"]):
        e = int(t["tor_bond"][k]); u, v = int(t["bond_src"][e]), int(t["bond_dst"][e])
        g = int(torch.searchsorted(t["lig_ptr"].long(), torch.tensor(u), right=True)) - 1
        l0 = int(t["lig_ptr"][g]); nl = int(t["lig_ptr"][g + 1]) - l0
        row = t["rot_mask"][int(t["rot_mask_off"][k]): int(t["rot_mask_off"][k]) + nl]
        assert row[v - l0] == 1 and row[u - l0] == 0
    slot = t["atom14_slot"]
    assert int((slot >= 0).sum()) == dm["NA"] and int(slot.max()) == dm["NA"] - 1
    # sc bonds are pocket-atom indices of the right graph
    assert int(t["sc_bond"].max()) < dm["NA"]
    assert dm["max_nl"] <= 256 and dm["max_na"] <= 2048


def test_synthetic_shapes_follow_the_data_contract():
    d = synthetic.make_batch(5, n_complex=1, poses=2, seed=1)
    NA, NL = d.rec_atm_pos.shape[0], d.lig_pos.shape[0]
    assert 2 * 0.85 * 600 < NA < 2 * 1.15 * 600 + 30 and 2 * 0.8 * 80 <= NL <= 2 * 1.2 * 80
    assert d.pocket_node_feature.shape == (NA, 5) and d.lig_node.shape == (NL, 27) and d.lig_edge_feat.shape[1] == 10
    assert d.default_frame.shape[1:] == (8, 4, 4) and d.rigid_group_positions.shape[1:] == (14, 3)
    assert int(d.atom14_mask.sum()) == NA and d.torsion_edge_index.shape[1:] == (4, 2)
    a37 = d.pocket_node_feature[:, 0].long()
    assert int(((a37 == 1) | (a37 == 3)).sum()) > 0
    # ligand radius graph must not hit torch_cluster's 32-neighbour cap
    from oracle.cluster import radius_graph
    ei = radius_graph(d.lig_pos, 5.0, d.lig_node_batch, max_num_neighbors=1000)
    assert torch.bincount(ei[1]).max() <= 32


def test_shard_lpt_is_balanced_and_complete():
    costs = [ddist.complex_cost(180 + 7 * (i % 9), 25 + (i % 11)) for i in range(128)]
    parts = ddist.shard_lpt(costs, 8)
    assert sorted(sum(parts, [])) == list(range(128))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) / min(loads) < 1.02


def test_shard_lpt_weighs_the_ranks_by_speed():
    """`shard_lpt(costs, world, speeds)`: boards hold clocks a few per cent apart at the power cap; a job goes to the rank that would finish
    it first, so the finish times (load / speed) even out instead of the loads."""
    rng = np.random.default_rng(2)
    costs = rng.uniform(0.9, 1.1, 640).tolist()
    speeds = [1.05, 1.0, 0.95, 1.0]
    parts = ddist.shard_lpt(costs, 4, speeds)
    assert sorted(i for p in parts for i in p) == list(range(640))
    t = [sum(costs[i] for i in p) / v for p, v in zip(parts, speeds)]
    assert max(t) / min(t) < 1.01                               # all ranks end together ...
    even = ddist.shard_lpt(costs, 4)
    t_even = [sum(costs[i] for i in p) / v for p, v in zip(even, speeds)]
    assert max(t_even) / min(t_even) > 1.08 and max(t) < 0.96 * max(t_even)      # ... where the even shard waits 5 % for the slow board
    assert ddist.shard_lpt(costs, 4, [1.0] * 4) == even
    assert ddist.rank_speeds(3.0, "cpu") == [1.0]               # one rank: nothing to weigh



def test_header_is_plain_c(tmp_path):
    """include/dbfr.h must compile as C (no C++/torch/hip types in the ABI)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "dbfr.h"\nint main(void){ dbfr_batch b; dbfr_step s; (void)b; (void)s; return sizeof(dbfr_model_cfg) > 0 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                        "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """sizeof / field offsets of the ctypes mirrors == what a C compiler lays out for include/dbfr.h."""
    import ctypes as C
    import subprocess
    fields = {"dbfr_batch": ["G", "max_nr", "lig_ptr", "rot_mask_off", "sc_ptr"], "dbfr_step": ["t", "sc_gsdt"],
              "dbfr_model_cfg": ["ns", "emb_scale", "no_sc_torsion"], "dbfr_tensor": ["numel"], "dbfr_noise": ["z_sc"],
              "dbfr_pose_metrics_in": ["n_res", "lig_traj", "aatype", "n_perm", "perms", "heavy_mask", "center", "chi_bound"],
              "dbfr_pose_metrics_out": ["centroid", "delta_chi", "lig_rmsd"],
              "dbfr_pdb_topology": ["n_res", "aatype", "b_factors", "remark"]}
    body = "".join(f'printf("{s} %zu\\n", sizeof({s}));' + "".join(f'printf("{s}.{f} %zu\\n", offsetof({s},{f}));' for f in fs)
                   for s, fs in fields.items())
    src = tmp_path / "o.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dbfr.h"\nint main(void){' + body + 'return 0;}\n')
    exe = tmp_path / "o"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    mirror = {"dbfr_batch": L.Batch, "dbfr_step": L.Step, "dbfr_model_cfg": L.ModelCfg, "dbfr_tensor": L.Tensor, "dbfr_noise": L.Noise,
              "dbfr_pose_metrics_in": L.PoseMetricsIn, "dbfr_pose_metrics_out": L.PoseMetricsOut, "dbfr_pdb_topology": L.PdbTopology}
    for s, cls in mirror.items():
        assert int(out[s]) == C.sizeof(cls), s
        for f in fields[s]:
            assert int(out[f"{s}.{f}"]) == getattr(cls, f).offset, (s, f)


def test_device_residue_tables_match_host_tables():
    """csrc/residue_tables.inc (compiled into the kernels) and data/residue_tables.npz (host look-ups) come from the same
    generator run: every integer table must agree."""
    import re
    inc = open(os.path.join(ROOT, "diffbindfr_amd", "csrc", "residue_tables.inc")).read()
    T = synthetic.residue_tables()

    def table(name):
        m = re.search(r"RT_TABLE int " + name + r"((?:\[\d+\])+) = \{([^}]*)\};", inc)
        assert m, name
        shape = [int(x) for x in re.findall(r"\[(\d+)\]", m.group(1))]
        return np.array([int(x) for x in m.group(2).split(",")]).reshape(shape)

    assert np.array_equal(table("kAtom14ToGroup").reshape(21, 14), T["atom14_to_group"])
    assert np.array_equal(table("kChiAtoms14"), T["chi_atoms14"])
    assert np.array_equal(table("kChiMask"), T["chi_mask"].astype(int))
    assert np.array_equal(table("kAtom14Mask"), T["atom14_mask"].astype(int))
    assert np.array_equal(table("kAtom37ToAtom14"), T["atom37_to_atom14"])
    assert np.array_equal(table("kAtom14ToAtom37"), T["atom14_to_atom37"])
    assert np.array_equal(table("kAtom37Mask"), T["atom37_mask"].astype(int))
    assert np.array_equal(table("kChiPiPeriodic"), T["chi_pi_periodic"].astype(int))
    assert np.array_equal(table("kAtom14Swap"), T["atom14_swap"])
    names = re.search(r"kAtom37Names\[37\]\[5\] = \{([^}]*)\};", inc).group(1).replace('"', "").split(", ")
    assert names == [str(a) for a in T["atom37_names"]]
    res3 = re.search(r"kRestypeNames3\[21\]\[4\] = \{([^}]*)\};", inc).group(1).replace('"', "").split(", ")
    assert res3 == [str(a) for a in T["restype_names3"]]


def test_f1_f2_entry_points_have_no_cpu_path():
    from diffbindfr_amd import assemble, pocket
    z = np.load(os.path.join(ROOT, "tests", "golden", "pocket.npz"))
    with pytest.raises(L.DbfrError):
        pocket.extract_templates(torch.from_numpy(z["aatype"]), torch.from_numpy(z["atom14_position"]).float())
    rng = np.random.default_rng(0)
    rec = synthetic.make_record(synthetic.make_pocket(rng, 40), synthetic.make_ligand(rng, 6), rng)
    pb = assemble.assemble([assemble.ComplexRecord(rec)], 2, "cpu")          # packing itself is host/torch only
    with pytest.raises(L.DbfrError):
        assemble.init_poses(None, pb, dict(tor=torch.zeros(1), rot=torch.eye(3).repeat(2, 1, 1), tr=torch.zeros(2, 3),
                                           sc=torch.zeros(pb.dims["NR"], 4)))


def test_library_build_id_matches_the_sources():
    """The in-tree libdbfr.so carries the hash of the sources it was built from (diffbindfr_amd/build.py)."""
    from diffbindfr_amd import build
    assert L.load().dbfr_build_id().decode() == build.source_hash()


def test_f16_range_guard_measures_the_row_depth():
    """The range guard behind `dbfr_model_fallback_convs` (api.cpp pack_f16_tiles, host code): a run's rows share ONE power-of-two factor;
    a row whose largest |w| lies 2^d below the run's keeps 22 significant bits in its two fp16 pieces while d <= 17.  The packer reports
    the largest d of a run; rows at 2^-10 of the maximum pass, a row at 2^-19 (the 'six decades inside a run' case) does not, an all-zero
    row (a padded channel) does not count, and a bias that dwarfs its weights (the factor stops at |bias| 2^k < 2^48) counts as depth."""
    import ctypes as C
    lib = L.load()
    rng = np.random.default_rng(5)
    nt = 3

    def depth(frag, bias):
        d, ok = C.c_int32(), C.c_int32()
        frag, bias = np.ascontiguousarray(frag, np.float32), np.ascontiguousarray(bias, np.float32)
        L.check(lib.dbfr_test_pack_f16_depth(frag.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p), nt, C.byref(d), C.byref(ok)))
        assert ok.value == 17
        return d.value

    base = rng.uniform(0.5, 1.0, (nt, 9, 64, 4)).astype(np.float32) * rng.choice([-1.0, 1.0], (nt, 9, 64, 4)).astype(np.float32)
    bias = rng.standard_normal((nt, 16)).astype(np.float32)
    assert depth(base, bias) <= 1                                   # rows of one size
    f = base.copy(); f[1, :, 5::16, :] *= 2.0 ** -10                # row 5 of tile 1 (lanes 5, 21, 37, 53 hold it): 2^-10 of the others
    assert 10 <= depth(f, bias) <= 11
    f = base.copy(); f[2, :, 9::16, :] *= 2.0 ** -19
    assert 19 <= depth(f, bias) <= 20 and depth(f, bias) > 17       # -> the conv would leave the fp16 kernel
    f = base.copy(); f[0, :, 3::16, :] = 0.0
    assert depth(f, bias) <= 1                                      # a padded (all-zero) row is not a deep row
    assert depth(base * 1e-30, bias) > 17                           # weights 1e-30 of their bias: the factor is held back by the bias rows


@pytest.mark.parametrize("scale", [1.0, 1e-4, 3e3])
def test_f16_tile_packer_scales_and_splits_exactly(scale):
    """DBFR_GEMM_SPLIT_F16's host side (api.cpp pack_f16_tiles through its test hook, no GPU): the tiles of a run are multiplied by ONE
    power of two that puts their largest magnitude into [2^14, 2^15); hi = fp16(v), lo = fp16(v - hi) then hold v to 2^-22 relative for
    everything within 2^-17 of the maximum and to 2^-25 absolute below (fp16 subnormals); the last 16 k sit as [hi 4 | lo 4] per lane
    (the A operand of the one x32 MFMA that carries both small products); the bias rows carry the same factor."""
    import ctypes as C
    lib = L.load()
    rng = np.random.default_rng(3)
    nt = 5
    frag = (rng.standard_normal((nt, 9, 64, 4)) * scale).astype(np.float32)
    frag[1] *= 1e-3                                          # a tile of much smaller rows inside the run
    frag[2, :, :, 0] = 0.0
    bias = (rng.standard_normal((nt, 16)) * scale).astype(np.float32)
    out = np.zeros(nt * 9280, np.uint8)
    k = C.c_int32()
    L.check(lib.dbfr_test_pack_f16_tiles(frag.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p), nt, out.ctypes.data_as(C.c_void_p), C.byref(k)))
    s = np.float64(2.0) ** k.value
    mx = np.abs(frag).max() * s
    assert 2 ** 14 <= mx < 2 ** 15
    tiles = out.reshape(nt, 9280)
    main = tiles[:, :8192].copy().view(np.float16).reshape(nt, 2, 4, 64, 8).astype(np.float64)       # [piece][k-step of 32][lane][8]
    tail = tiles[:, 8192:9216].copy().view(np.float16).reshape(nt, 64, 2, 4).astype(np.float64)       # [lane][hi | lo][4]
    b = tiles[:, 9216:9280].copy().view(np.float32).reshape(nt, 16)
    assert np.array_equal(b, (bias.astype(np.float64) * s).astype(np.float32))
    want = frag.astype(np.float64) * s                                                                   # [t][s4][lane][q]
    got = np.empty_like(want)
    for s4 in range(8):
        got[:, s4] = main[:, 0, s4 >> 1, :, 4 * (s4 & 1):4 * (s4 & 1) + 4] + main[:, 1, s4 >> 1, :, 4 * (s4 & 1):4 * (s4 & 1) + 4]
    got[:, 8] = tail[:, :, 0] + tail[:, :, 1]
    err = np.abs(got - want)
    assert (err <= np.maximum(np.abs(want) * 2.0 ** -22, 2.0 ** -25)).all(), float((err / np.maximum(np.abs(want), 1e-30)).max())
    assert np.isfinite(main).all() and np.abs(main[:, 0]).max() < 2 ** 15
    with pytest.raises(Exception):
        L.check(lib.dbfr_test_pack_f16_tiles(None, None, 0, None, None))


def test_f16_tile_packer_with_per_row_factors():
    """ABI 5: rows that lie further apart than two fp16 pieces hold behind one factor (here: every row its own size, log-uniform over TEN decades,
    a tile of all-zero rows, a row whose bias dwarfs it) are packed with one power of two per row: every row's largest magnitude lands in
    [2^14, 2^15), the pieces hold every value to 2^-22 of its ROW's maximum, rinv is the exact inverse power of two of the row's extra factor,
    the bias rows carry the row's factor too, and no depth is left -- except where the bias bound stops the factor."""
    import ctypes as C
    lib = L.load()
    rng = np.random.default_rng(9)
    nt = 4
    rowmag = 10.0 ** rng.uniform(-10, 0, (nt, 16))                     # [tile][row]
    frag = rng.standard_normal((nt, 9, 64, 4)).astype(np.float64)
    lane_row = np.arange(64) & 15
    frag *= rowmag[:, None, lane_row, None]
    frag[3] = 0.0                                                      # padded channels
    frag = frag.astype(np.float32)
    bias = (rng.standard_normal((nt, 16)) * rowmag).astype(np.float32)
    out = np.zeros(nt * 9280, np.uint8)
    k, depth = C.c_int32(), C.c_int32()
    rinv = np.zeros(nt * 16, np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    L.check(lib.dbfr_test_pack_f16_rows(vp(frag), vp(bias), nt, vp(out), C.byref(k), vp(rinv), C.byref(depth)))
    assert depth.value <= 0
    rinv = rinv.reshape(nt, 16).astype(np.float64)
    assert np.array_equal(np.log2(rinv), np.round(np.log2(rinv))) and (rinv <= 1).all() and (rinv[3] == 1).all()
    tiles = out.reshape(nt, 9280)
    main = tiles[:, :8192].copy().view(np.float16).reshape(nt, 2, 4, 64, 8).astype(np.float64)
    tail = tiles[:, 8192:9216].copy().view(np.float16).reshape(nt, 64, 2, 4).astype(np.float64)
    b = tiles[:, 9216:9280].copy().view(np.float32).reshape(nt, 16).astype(np.float64)
    got = np.empty((nt, 9, 64, 4))
    for s4 in range(8):
        got[:, s4] = main[:, 0, s4 >> 1, :, 4 * (s4 & 1):4 * (s4 & 1) + 4] + main[:, 1, s4 >> 1, :, 4 * (s4 & 1):4 * (s4 & 1) + 4]
    got[:, 8] = tail[:, :, 0] + tail[:, :, 1]
    rowmax = np.abs(got).reshape(nt, 9, 4, 16, 4).max(axis=(1, 2, 4))                                 # per (tile, row) after scaling
    assert ((rowmax[:3] >= 2 ** 14) & (rowmax[:3] < 2 ** 15)).all()
    s = 2.0 ** -k.value
    back = got * s * rinv[:, None, lane_row, None]                                                      # what the kernel's arithmetic sees
    want = frag.astype(np.float64)
    rm = np.abs(want).reshape(nt, 9, 4, 16, 4).max(axis=(1, 2, 4))[:, None, lane_row, None]
    assert (np.abs(back - want) <= np.maximum(np.abs(want) * 2.0 ** -22, rm * 2.0 ** -39)).all()
    assert np.allclose(b * s * rinv, bias.astype(np.float64), rtol=1e-7, atol=0)
    # a bias that dwarfs its row stops the row's factor (|bias| 2^(k + d) < 2^48): depth is left and reported
    bias2 = bias.copy(); bias2[0, 0] = 1e30
    L.check(lib.dbfr_test_pack_f16_rows(vp(frag), vp(bias2), nt, vp(out), C.byref(k), vp(rinv.astype(np.float32).ravel().copy()), C.byref(depth)))
    b2 = out.reshape(nt, 9280)[:, 9216:9280].copy().view(np.float32)
    assert np.isfinite(b2).all() and np.abs(b2).max() < 2.0 ** 48


def test_f16_tile_packer_keeps_the_bias_finite_next_to_tiny_weights():
    """A run of vanishing weights next to an O(1) bias: the factor stops where |bias| 2^k reaches 2^48, so that the bias times the
    kernel's per-edge factor (<= 2^64) times an input feature is still a finite fp32 number; an all-zero run gets the factor 1."""
    import ctypes as C
    lib = L.load()
    rng = np.random.default_rng(4)
    nt = 2
    frag = (rng.standard_normal((nt, 9, 64, 4)) * 1e-30).astype(np.float32)
    bias = rng.standard_normal((nt, 16)).astype(np.float32) * 3.0
    out = np.zeros(nt * 9280, np.uint8)
    k = C.c_int32()
    L.check(lib.dbfr_test_pack_f16_tiles(frag.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p), nt, out.ctypes.data_as(C.c_void_p), C.byref(k)))
    b = out.reshape(nt, 9280)[:, 9216:9280].copy().view(np.float32)
    assert 2.0 ** 47 <= np.abs(b).max() < 2.0 ** 48 and np.isfinite(b * np.float32(2.0 ** 64)).all()
    assert np.array_equal(b.reshape(nt, 16), (bias.astype(np.float64) * 2.0 ** k.value).astype(np.float32))
    frag[:] = 0
    L.check(lib.dbfr_test_pack_f16_tiles(frag.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p), nt, out.ctypes.data_as(C.c_void_p), C.byref(k)))
    assert k.value == 0


def test_register_into_druglib_with_a_stand_in_package(monkeypatch):
    """`register_into_druglib` against a stand-in `druglib` (the real one needs its whole environment): our classes land in ITS registries under
    their names, an existing entry of the same name is overwritten, and `DiffBindFRHIP` becomes a virtual subclass of the reference's ABCMeta base
    `BaseMLDocker` (druglib/models/Docking/base.py:13), so `isinstance` gates in a user's fork hold."""
    import sys, types
    from abc import ABCMeta
    from diffbindfr_amd import registry as R
    mods = {n: types.ModuleType(n) for n in ("druglib", "druglib.models", "druglib.models.builder", "druglib.models.Docking", "druglib.models.Docking.base")}
    b = mods["druglib.models.builder"]
    b.INTERACTION, b.MLDOCK_BUILDER, b.ENERGY = R.Registry("interaction"), R.Registry("mldock"), R.Registry("energy")

    class Old:                                                  # something a fork registered under our name before
        pass
    b.MLDOCK_BUILDER.register_module(name="DiffBindFRHIP", module=Old)

    class BaseMLDocker(metaclass=ABCMeta):
        pass
    mods["druglib.models.Docking.base"].BaseMLDocker = BaseMLDocker
    for n, m in mods.items():
        monkeypatch.setitem(sys.modules, n, m)
    assert not issubclass(dba.DiffBindFRHIP, BaseMLDocker)
    assert R.register_into_druglib() is True
    assert b.MLDOCK_BUILDER.get("DiffBindFRHIP") is dba.DiffBindFRHIP
    assert b.INTERACTION.get("TensorProductModelHIP") is dba.TensorProductModelHIP
    assert issubclass(dba.DiffBindFRHIP, BaseMLDocker) and isinstance(dba.DiffBindFRHIP(), BaseMLDocker)
