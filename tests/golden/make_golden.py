"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE'S OWN SOURCE.

Runs only in the build container (needs /root/reference; see ref_shims.py).
For every piece of the hot path whose reference source executes here it

  1. runs the reference function on seeded inputs,
  2. asserts the oracle restatement agrees (this is the oracle's pin), and
  3. freezes inputs + reference outputs as small .npz fixtures that travel to
     the GPU box, where tests compare oracle AND HIP path against them.

e3nn / torch_cluster / torch_scatter calls inside the reference are served by
the oracle's restatements (parity UNPINNED at that boundary, see oracle/__init__).

    python tests/golden/make_golden.py
"""
import copy
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_shims  # noqa: E402

ns = ref_shims.load_hot_path()
from oracle import geometry, sampler, schedule, score_model as sm  # noqa: E402
from diffbindfr_amd import synthetic  # noqa: E402

ED = ns.EasyDict
T = synthetic.residue_tables()


def npy(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def close(a, b, tol, what):
    a, b = torch.as_tensor(npy(a)).double(), torch.as_tensor(npy(b)).double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol, f"{what}: oracle vs reference max abs err {err} > {tol}"
    print(f"  pinned {what:45s} max|d|={err:.2e}")


def params_digest(params):
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(k.encode())
        h.update(npy(params[k]).tobytes())
    return h.hexdigest()


def ref_model_cfg():
    return ED(dict(
        task="struct_gen", no_sc_torsion=False,
        features_dim={"protein_atom": {"feature_list": ((37, 22, 4, 21, 2), 0)},
                      "ligand_atom": {"node_features": 27, "edge_features": 10}},
        ns=48, nv=12, sh_lmax=2, lig_cutoff=5, atom_cutoff=4, cross_cutoff=32, dynamic_max_cross=True,
        center_max_distance=32, atom_max_neighbors=1000, distance_embed_dim=32, time_emb_type="sinusoidal",
        sigma_embed_dim=32, emb_scale=1000, num_conv_layers=6, use_second_order_repr=False, dropout=0.1,
        batch_norm=True, scale_by_sigma=True))


def batch_to_npz(d):
    out = {k: npy(v) for k, v in vars(d).items() if torch.is_tensor(v)}
    for g, m in enumerate(d.rot_node_mask):
        out[f"rot_node_mask_{g}"] = npy(m)
    out["num_graphs"] = np.asarray(d.num_graphs)
    return out


# --------------------------------------------------------------------------- 1. geometry
def golden_geometry():
    print("[geometry]")
    g = torch.Generator().manual_seed(11)
    out = {}
    aa = torch.randn(16, 3, generator=g)
    aa[0] = torch.tensor([1e-8, 0.0, 0.0])           # small-angle branch
    aa[1] = torch.tensor([0.0, 0.0, 0.0]) + 3e-7
    R_ref = ns.geom.axis_angle_to_rot(aa)
    close(geometry.axis_angle_to_rot(aa), R_ref, 1e-6, "axis_angle_to_rot")
    out["aa"], out["aa_rot"] = npy(aa), npy(R_ref)

    A = torch.randn(3, 12, generator=g)
    Rt = ns.geom.axis_angle_to_rot(torch.tensor([0.3, -1.1, 0.7]))
    B = Rt @ A + torch.tensor([[1.0], [2.0], [-0.5]]) + 0.05 * torch.randn(3, 12, generator=g)
    R_ref, t_ref = ns.superimposition.rigid_transform_Kabsch_3D_torch(A, B)
    R_o, t_o = geometry.kabsch(A, B)
    close(R_o, R_ref, 1e-6, "kabsch R")
    close(t_o, t_ref, 1e-6, "kabsch t")
    out.update(kabsch_A=npy(A), kabsch_B=npy(B), kabsch_R=npy(R_ref), kabsch_t=npy(t_ref))

    # update_batchlig_pos: 3 ligands incl. one with no torsion (tiny ring-only ligand)
    rng = np.random.default_rng(5)
    ligs = [synthetic.make_ligand(rng, 14), synthetic.make_ligand(rng, 9), synthetic.make_ligand(rng, 4)]
    ligs[2]["tor_edge_mask"][:] = False
    ligs[2]["rot_node_mask"] = np.zeros((0, 4), bool)
    pk = synthetic.make_pocket(rng, 30)
    items = [(pk, lg) + synthetic.init_pose(rng, pk, lg) for lg in ligs]
    d = synthetic.collate(items)
    n_tor = int(d.tor_edge_mask.sum())
    tr, rot = torch.randn(3, 3, generator=g), 0.5 * torch.randn(3, 3, generator=g)
    tor = torch.randn(n_tor, generator=g)
    tor[1] = 0.0                                        # exercises the ==0 skip
    ref = ns.conformer.update_batchlig_pos(tr, rot, tor, d.lig_pos, d.lig_edge_index, d.tor_edge_mask,
                                           d.rot_node_mask, batch=d.lig_node_batch)
    close(geometry.update_batchlig_pos(tr, rot, tor, d.lig_pos, d.lig_edge_index, d.tor_edge_mask,
                                       d.rot_node_mask, d.lig_node_batch), ref, 2e-5, "update_batchlig_pos")
    out.update({f"lig_{k}": v for k, v in batch_to_npz(d).items()
                if k.startswith(("lig_", "tor_edge", "rot_node"))})
    out.update(lig_tr=npy(tr), lig_rot=npy(rot), lig_tor=npy(tor), lig_new_pos=npy(ref))

    # build_pdb_from_template: all 20 residue types, perturbed templates, random chi
    seq = torch.arange(21) % 20
    N = len(seq)
    transl = 5 * torch.randn(N, 3, generator=g)
    rots = ns.geom.axis_angle_to_rot(torch.randn(N, 3, generator=g))
    dframe = torch.from_numpy(T["default_frame"])[seq].clone()
    rigid = torch.from_numpy(T["atom14_lit_pos"])[seq] + 0.05 * torch.randn(N, 14, 3, generator=g)
    ang = (torch.rand(N, 5, generator=g) * 2 - 1) * np.pi
    tmpl = ED(sequence=seq, backbone_transl=transl, backbone_rots=rots, default_frame=dframe,
              rigid_group_positions=rigid, torsion_angle=ns.geom.radian2sincos_torch(ang))
    ref14, _ = ns.prot_math.build_pdb_from_template(tmpl, torch.device("cpu"))
    o14 = geometry.build_atom14(seq, transl, rots, dframe, rigid, ang, torch.from_numpy(T["atom14_to_group"]).long())
    close(o14, ref14, 2e-5, "build_pdb_from_template")
    out.update(sc_seq=npy(seq), sc_transl=npy(transl), sc_rots=npy(rots), sc_default_frame=npy(dframe),
               sc_rigid=npy(rigid), sc_angle=npy(ang), sc_atom14=npy(ref14))
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **out)


# --------------------------------------------------------------------------- 2. embeddings
def golden_embeddings():
    print("[embeddings]")
    g = torch.Generator().manual_seed(12)
    out = {}
    t = torch.tensor([1.0, 0.545459, 0.136372, 1e-5])
    ref = ns.time_emb.sinusoidal_embedding(1000 * t, 32)
    close(sm.sinusoidal_embedding(1000 * t, 32), ref, 0, "sinusoidal_embedding")
    out.update(temb_t=npy(t), temb=npy(ref))
    for stop in (4.0, 5.0, 32.0):
        gs = ns.schnet.GaussianSmearing(0.0, stop, 32)
        dist = torch.rand(40, generator=g) * stop * 1.3
        ref = gs(dist)
        p = {"x.offset": gs.offset, "x.coeff": gs.coeff}
        close(sm.gaussian_smearing(p, "x", dist), ref, 0, f"GaussianSmearing stop={stop}")
        out[f"gs{int(stop)}_d"], out[f"gs{int(stop)}"] = npy(dist), npy(ref)
    ei, ne = ns.torch_utils.get_complete_bipartite_graph(torch.tensor([2, 3, 2]), torch.tensor([4, 2, 5]))
    close(sm.complete_bipartite(torch.tensor([2, 3, 2]), torch.tensor([4, 2, 5])), ei, 0, "complete_bipartite")
    out["bip"] = npy(ei)
    irreps = "48x0e + 12x1o + 12x1e + 48x0o"
    ln = ns.tpscore.LayerNorm(irreps)
    with torch.no_grad():
        ln.mean_shift += 0.1 * torch.randn(ln.mean_shift.shape, generator=g)
        ln.affine_weight += 0.1 * torch.randn(ln.affine_weight.shape, generator=g)
        ln.affine_bias += 0.1 * torch.randn(ln.affine_bias.shape, generator=g)
        x = torch.randn(7, 168, generator=g)
        x[3] = 0                                        # isolated node -> all-zero row
        ref = ln(x)
    p = {"n.mean_shift": ln.mean_shift.detach(), "n.affine_weight": ln.affine_weight.detach(),
         "n.affine_bias": ln.affine_bias.detach()}
    close(sm.layer_norm(p, "n", irreps, x), ref, 0, "equivariant LayerNorm")
    out.update(ln_x=npy(x), ln_y=npy(ref), ln_mean_shift=npy(p["n.mean_shift"]),
               ln_weight=npy(p["n.affine_weight"]), ln_bias=npy(p["n.affine_bias"]))
    np.savez_compressed(os.path.join(HERE, "embeddings.npz"), **out)


# --------------------------------------------------------------------------- 3. schedule
def _exec_prefix(path, stop_marker):
    src = open(path).read()
    src = src[: src.index(stop_marker)].replace("from . import io", "")
    env = {"__name__": "ref_prefix"}
    exec(compile(src, path, "exec"), env)
    return env


def golden_schedule():
    print("[schedule]")
    out = {}
    scfg = schedule.default_sample_cfg()
    test_cfg = ED(sample_cfg=ED(vars(scfg)))
    sampler_ref = ns.scflex.DiffBindFR(diffusion_model=None, test_cfg=test_cfg)
    sampler_ref.diffusion_model_cfg = ED(no_sc_torsion=False)
    ts = sampler_ref.t_schedule()
    close(schedule.t_schedule(scfg), ts, 0, "t_schedule")
    rows = []
    for i in range(scfg.actual_steps):
        t, dt = ts[i], ts[i] - ts[i + 1]
        s = sampler_ref.sigma_fn(t, t, t, t)
        sc = schedule.step_scalars(scfg, i)
        for a, b in zip(s, (sc.tr_sigma, sc.rot_sigma, sc.tor_sigma, sc.sc_tor_sigma)):
            assert float(a) == float(b)
        rows.append([float(t), float(dt)] + [float(x) for x in s] + [float(sc.tr_g), float(sc.rot_g), float(sc.tor_g)])
    out["steps"] = np.asarray(rows, np.float64)
    # SURVEY Appendix B.4 known-answer spot checks
    assert abs(rows[0][2] - 6.0) < 1e-5 and abs(rows[19][2] - 0.17478) < 1e-4 and abs(rows[0][6] - 17.16953) < 1e-3
    so3 = _exec_prefix(os.path.join(ref_shims.COPY, "druglib/utils/geometry_utils/so3.py"), "resource_dir =")
    eps_arr = 10 ** np.linspace(np.log10(so3["MIN_EPS"]), np.log10(so3["MAX_EPS"]), so3["N_EPS"])
    om = np.linspace(0, np.pi, so3["X_N"] + 1)[1:]
    idxs, vals = [952, 613, 309, 0, 999], []
    for i in idxs:
        ex = so3["_expansion"](om, eps_arr[i])
        pdf = so3["_density"](ex, om, marginal=True)
        scn = so3["_score"](ex, om, eps_arr[i])
        vals.append(np.sqrt(np.sum(scn ** 2 * pdf) / np.sum(pdf) / np.pi))
        close(schedule.so3_exp_score_norm(i), vals[-1], 1e-12 * max(1, abs(vals[-1])), f"so3 exp_score_norm[{i}]")
    out["so3_idx"], out["so3_val"] = np.asarray(idxs), np.asarray(vals)
    sig = np.array([1.55, 0.25799, 0.05138], np.float32)
    assert list(schedule.so3_eps_index(sig)) == [952, 613, 309]
    tor = _exec_prefix(os.path.join(ref_shims.COPY, "druglib/utils/geometry_utils/torus.py"), "resource_dir =")
    tor["tqdm"] = types.SimpleNamespace(trange=range)
    tidx = [4467, 2857, 1408, 0, 5000]
    rows_ref = []
    for i in tidx:
        sg = tor["sigma"][i:i + 1]
        p_ = tor["p"](tor["x"], sg[:, None], N=100)
        sc_ = tor["grad"](tor["x"], sg[:, None], N=100) / p_
        rows_ref.append(sc_[0])
        # oracle's row (recomputed the same way inside torus_score_norm_entry) -> compare via the MC mean
        np.random.seed(1234)
        smp = sg[0] * np.random.randn(200000)
        smp = (smp + np.pi) % (2 * np.pi) - np.pi
        xs = np.log(np.abs(smp) / np.pi)
        xs = np.round(np.clip((xs - np.log(tor["X_MIN"])) / (0 - np.log(tor["X_MIN"])) * tor["X_N"], 0, tor["X_N"])).astype(int)
        big = float((sc_[0][xs] ** 2).mean())
        mine = schedule.torus_score_norm_entry(i, seed=0)
        assert abs(mine - big) / big < 0.08, (i, mine, big)
        print(f"  pinned torus score_norm_[{i}] oracle(10k draws)={mine:.5f} ref-table(200k draws)={big:.5f}")
    assert list(schedule.torus_sigma_index(np.array([3.14, 0.38712, 0.05884], np.float32))) == [4467, 2857, 1408]
    out["torus_idx"] = np.asarray(tidx)
    out["torus_score_rows_head"] = np.asarray([r[::250] for r in rows_ref])
    out["torus_norm_seed0"] = np.asarray([schedule.torus_score_norm_entry(i, 0) for i in tidx])
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), **out)


# --------------------------------------------------------------------------- 4/5. model + sampler
def golden_model_and_sampler():
    print("[score model + sampler]")
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=1)
    digest = params_digest(params)
    model = ns.tpscore.TensorProductModel(ref_model_cfg()).eval()
    sd = model.state_dict()
    assert sorted(sd) == sorted(params), "state_dict key set differs from the reference module"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(params[k].shape), k
    model.load_state_dict(params, strict=True)
    scfg = schedule.default_sample_cfg()
    d = synthetic.make_batch(2, n_complex=2, poses=2, seed=3, n_atoms=60, n_lig=10)
    G = d.num_graphs
    out = batch_to_npz(d)
    out["params_seed"], out["params_sha256"] = np.asarray(1), np.asarray(digest)
    for step in (0, 10, 19):
        sc = schedule.step_scalars(scfg, step)
        with torch.no_grad():
            ref = model(ED(vars(sampler.set_time(copy.deepcopy(d), sc, G))))
            mine = sm.forward(params, mcfg, sampler.set_time(copy.deepcopy(d), sc, G))
        for nm, a, b in zip(("tr", "rot", "tor", "sc_tor"), mine, ref):
            close(a, b, 1e-6, f"score[{nm}] step {step}")
            out[f"score_{nm}_{step}"] = npy(b)

    # sampler: reference DiffBindFR.sample with the so3/torus tables served by the oracle
    # (their own import would build GB-sized tables); noise from torch's global generator.
    go = sys.modules["druglib.utils.geometry_utils"]
    go.so3 = types.SimpleNamespace(score_norm=schedule.so3_score_norm)
    go.torus = types.SimpleNamespace(score_norm=lambda s: schedule.torus_score_norm(s, 0))
    test_cfg = ED(sample_cfg=ED(vars(scfg)))
    ref_sampler = ns.scflex.DiffBindFR(diffusion_model=None, test_cfg=test_cfg)
    ref_sampler.diffusion_model_cfg = ED(no_sc_torsion=False)
    ref_sampler.diffusion_model = model
    rd = ED({k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(d).items() if k != "rot_node_mask"})
    rd.metastore = {"rot_node_mask": [m.clone() for m in d.rot_node_mask]}
    torch.manual_seed(77)
    res = ref_sampler.sample(rd, visualize=True)
    lig_ref = torch.cat([r[0] for r in res], dim=1)            # [T, N_l_total, 3]
    a14_ref = torch.cat([r[1] for r in res], dim=1)            # [T, N_res_total, 14, 3]
    n_tor, n_sc = int(d.tor_edge_mask.sum()), int(d.sc_torsion_edge_mask.sum())
    noise = sampler.draw_noise(scfg.actual_steps, G, n_tor, n_sc, seed=77)
    lig_o, a14_o = sampler.sample(params, mcfg, scfg, copy.deepcopy(d), noise,
                                  torch.from_numpy(T["atom14_to_group"]).long(), torus_seed=0, visualize=True)
    close(lig_o, lig_ref, 1e-4, "sample(): ligand trajectory (20 steps)")
    close(a14_o, a14_ref, 1e-4, "sample(): atom14 trajectory (20 steps)")
    out.update(noise_seed=np.asarray(77), noise_tr=npy(noise.tr), noise_rot=npy(noise.rot), noise_tor=npy(noise.tor),
               noise_sc=npy(noise.sc), traj_lig=npy(lig_ref), traj_atom14=npy(a14_ref))
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)


def golden_sampler_modes():
    """The sampler's other branches on the small batch of sampler.npz (inputs are there): `sample_cfg.type = 'ode'` (scFlex.py:162-165,
    199-200: 0.5 g^2 score dt, nothing drawn) and `no_random = True` (:167-183: SDE drift, z = 0) through the reference's own
    DiffBindFR.sample(); the oracle must reproduce both; `time_schedule != 'linear'` must raise (:91).  -> tests/golden/sampler_modes.npz"""
    print("[sampler modes: ode / no_random / time_schedule guard]")
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=1)
    model = ns.tpscore.TensorProductModel(ref_model_cfg()).eval()
    model.load_state_dict(params, strict=True)
    d = synthetic.make_batch(2, n_complex=2, poses=2, seed=3, n_atoms=60, n_lig=10)
    G = d.num_graphs
    go = sys.modules["druglib.utils.geometry_utils"]
    go.so3 = types.SimpleNamespace(score_norm=schedule.so3_score_norm)
    go.torus = types.SimpleNamespace(score_norm=lambda s: schedule.torus_score_norm(s, 0))
    n_tor, n_sc = int(d.tor_edge_mask.sum()), int(d.sc_torsion_edge_mask.sum())
    out = {}
    for tag, over in (("ode", dict(type="ode")), ("no_random", dict(no_random=True))):
        scfg = schedule.default_sample_cfg(**over)
        ref_sampler = ns.scflex.DiffBindFR(diffusion_model=None, test_cfg=ED(sample_cfg=ED(vars(scfg))))
        ref_sampler.diffusion_model_cfg = ED(no_sc_torsion=False)
        ref_sampler.diffusion_model = model
        rd = ED({k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(d).items() if k != "rot_node_mask"})
        rd.metastore = {"rot_node_mask": [m.clone() for m in d.rot_node_mask]}
        torch.manual_seed(5)
        state = torch.get_rng_state()
        res = ref_sampler.sample(rd, visualize=True)
        assert torch.equal(torch.get_rng_state(), state), f"{tag}: the reference drew random numbers"
        lig_ref = torch.cat([r[0] for r in res], dim=1)
        a14_ref = torch.cat([r[1] for r in res], dim=1)
        noise = sampler.draw_noise(scfg.actual_steps, G, n_tor, n_sc, seed=5, none=True)
        lig_o, a14_o = sampler.sample(params, mcfg, scfg, copy.deepcopy(d), noise, torch.from_numpy(T["atom14_to_group"]).long(),
                                      torus_seed=0, visualize=True)
        close(lig_o, lig_ref, 1e-4, f"sample() {tag}: ligand trajectory (20 steps)")
        close(a14_o, a14_ref, 1e-4, f"sample() {tag}: atom14 trajectory (20 steps)")
        out[f"{tag}_traj_lig"], out[f"{tag}_traj_atom14"] = npy(lig_ref), npy(a14_ref)
    z = np.load(os.path.join(HERE, "sampler.npz"))
    dsde = float(np.abs(z["traj_lig"] - out["ode_traj_lig"]).max())
    print(f"  ode vs sde final ligand positions differ by up to {dsde:.2f} A (the modes are not interchangeable)")
    assert dsde > 0.1
    scfg = schedule.default_sample_cfg(time_schedule="cosine")
    ref_sampler = ns.scflex.DiffBindFR(diffusion_model=None, test_cfg=ED(sample_cfg=ED(vars(scfg))))
    try:
        ref_sampler.t_schedule()
        raise AssertionError("the reference accepted time_schedule='cosine'")
    except NotImplementedError as e:
        out["time_schedule_error"] = np.asarray(str(e))
    out["params_seed"] = np.asarray(1)
    np.savez_compressed(os.path.join(HERE, "sampler_modes.npz"), **out)
    print("  sampler_modes.npz:", os.path.getsize(os.path.join(HERE, "sampler_modes.npz")) // 1024, "KiB")


def golden_pose_init():
    """f1: LigInit / SCFixer / SCProtInit / Atom14ToAllAtomsRepr of the reference's struct_init.py run on seeded
    global generators; the oracle gets the same draws as an explicit tape."""
    print("[pose init]")
    from scipy.spatial.transform import Rotation
    from oracle import pose_init as opi
    dsb = types.ModuleType("druglib.datasets.builder")
    dsb.PIPELINES = ns.builder.INTERACTION.__class__("pipeline")
    sys.modules["druglib.datasets.builder"] = dsb
    sys.modules["druglib.utils.geometry_utils"].radian2sincos_torch = ns.geom.radian2sincos_torch
    sys.modules["druglib.utils"].get_logger = lambda name: __import__("logging").getLogger(name)
    # the reference's real data package (Data / DataContainer / Batch / collate); torch_sparse & PyG stay stubs
    class SparseTensor:
        pass
    sys.modules["torch_sparse"].SparseTensor = SparseTensor
    sys.modules["torch_sparse"].cat = lambda *a, **k: None
    for nm in ["torch_geometric.utils.num_nodes", "torch_geometric.utils.hetero", "torch_geometric.data",
               "torch_geometric.data.storage", "torch_geometric.data.data"]:
        sys.modules[nm] = ref_shims._Anything(nm)
    sys.modules["druglib.utils"].color = types.SimpleNamespace()
    for k in [k for k in sys.modules if k == "druglib.data" or k.startswith("druglib.data.")]:
        del sys.modules[k]
    D = __import__("importlib").import_module("druglib.data")
    si = ref_shims._load("druglib.datasets.Docking.struct_init", "datasets/Docking/struct_init.py")
    fm = ref_shims._load("druglib.datasets.Docking.formatting", "datasets/Docking/formatting.py")
    rng = np.random.default_rng(4711)
    cases = [("plain", synthetic.make_record(synthetic.make_pocket(rng, 60), synthetic.make_ligand(rng, 14), rng)),
             ("fixer", synthetic.make_record(synthetic.make_pocket(rng, 80), synthetic.make_ligand(rng, 22), rng, drop_sidechains=3)),
             ("rigid", synthetic.make_record(synthetic.make_pocket(rng, 40), synthetic.make_ligand(rng, 5), rng))]
    cases[2][1]["tor_edge_mask"].zero_()
    cases[2][1]["rot_node_mask"] = cases[2][1]["rot_node_mask"][:0]
    out = {}
    for name, rec in cases:
        for seed in (3, 4):
            d = copy.deepcopy({k: v for k, v in rec.items() if k != "rot_node_mask"})
            d["metastore"] = dict(rot_node_mask=rec["rot_node_mask"].clone())
            np.random.seed(seed)
            torch.manual_seed(seed)
            d = si.LigInit(tr_sigma_max=10)(d)
            d = si.SCFixer()(d)
            d = si.SCProtInit()(d)
            d = fm.Atom14ToAllAtomsRepr()(d)
            # the same draws, in the reference's order, as an explicit tape
            np.random.seed(seed)
            torch.manual_seed(seed)
            n_tor, n_res = int(rec["tor_edge_mask"].sum()), rec["sequence"].shape[0]
            tape = {}
            if n_tor:
                tape["tor"] = np.random.uniform(low=-np.pi, high=np.pi, size=n_tor)
            tape["rot"] = Rotation.random().as_matrix()
            tape["tr"] = torch.normal(mean=0, std=10., size=(1, 3))
            tape["sc"] = np.random.uniform(low=-np.pi, high=np.pi, size=(n_res, 4))
            fixed = opi.sc_fixer(copy.deepcopy(rec), T)
            mine = opi.init_pose(fixed, tape, {k: torch.from_numpy(np.asarray(v)) if k == "atom14_to_group" else v for k, v in T.items()})
            for k in ("lig_pos", "torsion_angle", "rec_atm_pos", "pocket_node_feature", "sc_torsion_edge_mask", "atom14_mask",
                      "default_frame", "rigid_group_positions"):
                close(mine[k].float(), d[k].float(), 0.0, f"{name}/s{seed}/{k}")
            pre = f"{name}_s{seed}_"
            out.update({pre + "tape_" + k: npy(v) for k, v in tape.items()})
            out.update({pre + "out_" + k: npy(d[k]) for k in ("lig_pos", "torsion_angle", "rec_atm_pos", "pocket_node_feature",
                                                                "sc_torsion_edge_mask", "atom14_mask", "default_frame",
                                                                "rigid_group_positions")})
        out.update({f"{name}_rec_{k}": npy(v) for k, v in rec.items()})
    # ---- collate: 2 complexes x 2 poses through the reference's Batch.from_data_list (follow_batch as diffbindfr_ts.py:92-96)
    Tt = {k: torch.from_numpy(np.asarray(v)) if k == "atom14_to_group" else v for k, v in T.items()}
    poses = []
    for name, rec in cases[:2]:
        fixed = opi.sc_fixer(copy.deepcopy(rec), T)
        for p in range(2):
            n_tor = int(rec["tor_edge_mask"].sum())
            tape = dict(tor=rng.uniform(-np.pi, np.pi, n_tor), rot=Rotation.random(random_state=7 + p).as_matrix(),
                        tr=torch.from_numpy(rng.normal(0, 10, (1, 3))).float(), sc=rng.uniform(-np.pi, np.pi, (rec["sequence"].shape[0], 4)))
            poses.append(opi.init_pose(fixed, tape, Tt))
            out.update({f"collate_tape{len(poses) - 1}_{k}": npy(v) for k, v in tape.items()})
    fields = ("lig_node", "lig_pos", "lig_edge_index", "lig_edge_feat", "tor_edge_mask", "pocket_node_feature", "rec_atm_pos",
              "sc_torsion_edge_mask", "torsion_edge_index", "backbone_transl", "sequence", "atom14_mask", "backbone_rots",
              "default_frame", "rigid_group_positions", "torsion_angle")
    items = []
    for p in poses:
        d = {k: D.DataContainer(p[k], stack=False, is_graph=True) for k in fields}
        d["metastore"] = dict(rot_node_mask=p["rot_node_mask"])
        items.append(fm.ToPLData()(d))
    ref = D.Batch.from_data_list(items, follow_batch=["lig_node", "rec_atm_pos"], exclude_keys=None).to_dict(decode=True, drop_meta=False)
    mine = opi.collate(poses)
    for k, v in ref.items():
        if torch.is_tensor(v) and k not in ("batch", "ptr"):
            assert v.dtype == mine[k].dtype and v.shape == mine[k].shape, k
            close(mine[k].double(), v.double(), 0.0, f"collate/{k}")
            out["collate_" + k] = npy(v)
    assert all(bool((a == b).all()) for a, b in zip(ref["metastore"]["rot_node_mask"], mine["rot_node_mask"]))
    np.savez_compressed(os.path.join(HERE, "pose_init.npz"), **out)


def golden_pocket():
    """f2: once-per-pocket preparation -- extract_chi_and_template, make_torsion_mask, build_torsion_edges,
    PocketFeaturizer, Decentration of the reference on a synthetic all-residue-type pocket (float64 coordinates,
    as Protein.atom_positions holds them)."""
    print("[pocket]")
    from oracle import pocket as opk
    if "druglib.datasets.builder" not in sys.modules or not hasattr(sys.modules["druglib.datasets.builder"], "PIPELINES"):
        dsb = types.ModuleType("druglib.datasets.builder")
        dsb.PIPELINES = ns.builder.INTERACTION.__class__("pipeline")
        sys.modules["druglib.datasets.builder"] = dsb
    obj = sys.modules["druglib.utils.obj"]
    obj.Ligand3D = object
    obj.make_torsion_mask = ns.prot_math.make_torsion_mask
    du = ref_shims._load("druglib.datasets.Docking.utils", "datasets/Docking/utils.py")
    pp = ref_shims._load("druglib.datasets.Docking.pocket_pipeline", "datasets/Docking/pocket_pipeline.py")
    rng = np.random.default_rng(808)
    # every residue type at least twice, random order
    seq = np.concatenate([np.arange(20), np.arange(20), rng.integers(0, 20, 24)])
    rng.shuffle(seq)
    N = len(seq)
    transl = rng.normal(0, 8.0, (N, 3))
    rots = np.stack([synthetic._rand_rot(rng) for _ in range(N)])
    m14 = T["atom14_mask"][seq].astype(bool)
    rigid = T["atom14_lit_pos"][seq].astype(np.float64) + rng.normal(0, 0.04, (N, 14, 3)) * m14[..., None]
    tor = rng.uniform(-np.pi, np.pi, (N, 5))
    pos = synthetic.build_atom14_np(seq, transl, rots, T["default_frame"][seq].astype(np.float64), rigid, tor, T["atom14_to_group"])
    ideal = T["atom14_mask"][seq][..., None].astype(np.float32)
    pos = pos * ideal                                            # float64, zeros in unused slots (to_pos14)
    ref = ns.prot_math.extract_chi_and_template(seq, pos.copy(), ideal, return_radian=True)
    mine = opk.extract_chi_and_template(seq, pos.copy(), ideal, T)
    out = dict(aatype=seq, atom14_position=pos, ideal_mask=ideal[..., 0])
    for k in ("backbone_transl", "backbone_rots", "default_frame", "rigid_group_positions", "torsion_angle"):
        close(mine[k], ref[k], 0.0, "pocket/" + k)
        out["ref_" + k] = npy(ref[k])
    # the reference's own inverse: templates -> atom14 must give the input back
    back = ns.prot_math.build_pdb_from_template(ED(sequence=torch.from_numpy(seq), backbone_transl=torch.from_numpy(ref["backbone_transl"]).float(),
                                                    backbone_rots=torch.from_numpy(ref["backbone_rots"]).float(),
                                                    default_frame=torch.from_numpy(ref["default_frame"]),
                                                    rigid_group_positions=torch.from_numpy(ref["rigid_group_positions"]),
                                                    torsion_angle=ns.geom.radian2sincos_torch(torch.from_numpy(ref["torsion_angle"]))),
                                                 torch.device("cpu"))[0]
    close(back * torch.from_numpy(ideal), torch.from_numpy(pos).float(), 2e-4, "pocket/round trip through build_pdb_from_template")
    # masks / edges / features with some side chains missing
    actual = torch.from_numpy(m14.copy())
    for r in rng.choice(np.nonzero(m14[:, 6:].any(1))[0], 6, replace=False):
        actual[r, 6:] = False
    actual[3, 5] = False
    seq_t = torch.from_numpy(seq)
    close(opk.make_torsion_mask(seq_t, actual, T).float(), ns.prot_math.make_torsion_mask(seq_t, actual).float(), 0.0, "pocket/make_torsion_mask")
    te_ref, cm_ref = du.build_torsion_edges(seq_t, actual)
    te, cm = opk.build_torsion_edges(seq_t, actual, T)
    close(te, te_ref[..., 1, :], 0.0, "pocket/torsion_edge_index")
    close(cm.float(), cm_ref.float(), 0.0, "pocket/sc_torsion_edge_mask")
    fake = types.SimpleNamespace(num_res=lambda: N, atom_mask=np.zeros((N, 37)), residue_prop={})
    d = pp.PocketFeaturizer()(dict(atom14_mask=actual, sequence=seq_t, pocket=fake))
    close(opk.pocket_features(seq_t, actual, T), d["pocket_node_feature"], 0.0, "pocket/pocket_node_feature")
    out.update(actual_mask=npy(actual), ref_torsion_edge_index=npy(te_ref[..., 1, :]), ref_sc_torsion_edge_mask=npy(cm_ref),
               ref_pocket_node_feature=npy(d["pocket_node_feature"]))
    np.savez_compressed(os.path.join(HERE, "pocket.npz"), **out)


def _parse_sdf_heavy(path):
    """V2000 mol block -> heavy-atom coordinates, elements, bonds (hydrogens dropped, order kept as RDKit's
    RemoveHs does)."""
    L = open(path).read().split("\n")
    na, nb = int(L[3][0:3]), int(L[3][3:6])
    xyz = [[float(L[4 + i][0:10]), float(L[4 + i][10:20]), float(L[4 + i][20:30])] for i in range(na)]
    el = [L[4 + i][31:34].strip() for i in range(na)]
    bonds = [(int(L[4 + na + i][0:3]) - 1, int(L[4 + na + i][3:6]) - 1) for i in range(nb)]
    keep = [i for i, e in enumerate(el) if e != "H"]
    ren = {a: k for k, a in enumerate(keep)}
    return np.array([xyz[i] for i in keep]), [(ren[a], ren[b]) for a, b in bonds if a in ren and b in ren]


def _parse_pdb_residues(path):
    """ATOM records -> [(key, resname, {atom name: xyz})] in file order; heavy atoms, first alternate location."""
    res, order = {}, []
    for l in open(path):
        if not l.startswith("ATOM") or l[16] not in " A":
            continue
        name, el = l[12:16].strip(), (l[76:78].strip() if len(l) > 77 else "")
        if el == "H" or (not el and name[0] == "H"):
            continue
        key = (l[21], int(l[22:26]), l[26])
        if key not in res:
            res[key] = (l[17:20], {})
            order.append(key)
        res[key][1].setdefault(name, [float(l[30:38]), float(l[38:46]), float(l[46:54])])
    return [(k,) + res[k] for k in order]


def golden_real_complex():
    """BASELINE config 1's structure: examples/forward/3dbs_protein.pdb + its crystal ligand, pocket = residues with any
    heavy atom within 12 A of any ligand heavy atom (diffbindfr_ts.py:33-41).  Real side-chain geometry through the
    reference's extract_chi_and_template / masks / edges, the real bond graph through its find_torsion."""
    print("[real complex 3DBS]")
    from oracle import pocket as opk, ligand as olig, cluster
    from diffbindfr_amd import ligand as plig
    obj = sys.modules["druglib.utils.obj"]
    obj.Ligand3D = object
    obj.make_torsion_mask = ns.prot_math.make_torsion_mask
    du = ref_shims._load("druglib.datasets.Docking.utils", "datasets/Docking/utils.py")
    ex = os.path.join(ref_shims.REF, "examples", "forward")
    lig, bonds = _parse_sdf_heavy(os.path.join(ex, "3dbs_protein_crystal.sdf"))
    pc = ns.pc
    seq, pos, msk = [], [], []
    for key, rn, atoms in _parse_pdb_residues(os.path.join(ex, "3dbs_protein.pdb")):
        if rn not in pc.restype_3to1:
            continue
        P = np.array(list(atoms.values()))
        if np.linalg.norm(P[:, None] - lig[None], axis=-1).min() >= 12.0:
            continue
        names = pc.restype_name_to_atom14_names[rn]
        seq.append(pc.restype_order[pc.restype_3to1[rn]])
        pos.append([atoms.get(nm, [0.0, 0.0, 0.0]) if nm else [0.0, 0.0, 0.0] for nm in names])
        msk.append([bool(nm) and nm in atoms for nm in names])
    seq, pos, msk = np.asarray(seq), np.asarray(pos, np.float64), np.asarray(msk)
    n_l = lig.shape[0]
    assert (len(seq), int(msk.sum()), int(msk[:, 1].sum() + msk[:, 4].sum()), n_l) == (105, 866, 205, 35), "SURVEY.md section 8 counts"
    ideal = T["atom14_mask"][seq][..., None].astype(np.float32)
    ref = ns.prot_math.extract_chi_and_template(seq, pos.copy(), ideal, return_radian=True)
    mine = opk.extract_chi_and_template(seq, pos.copy(), ideal, T)
    out = dict(aatype=seq, atom14_position=pos, atom14_mask=msk, lig_pos=lig)
    for k in ("backbone_transl", "backbone_rots", "default_frame", "rigid_group_positions", "torsion_angle"):
        close(mine[k], ref[k], 0.0, "3dbs/" + k)
        out["ref_" + k] = npy(ref[k])
    seq_t, msk_t = torch.from_numpy(seq), torch.from_numpy(msk)
    te_ref, cm_ref = du.build_torsion_edges(seq_t, msk_t)
    te, cm = opk.build_torsion_edges(seq_t, msk_t, T)
    close(te, te_ref[..., 1, :], 0.0, "3dbs/torsion_edge_index")
    close(cm.float(), cm_ref.float(), 0.0, "3dbs/sc_torsion_edge_mask")
    out.update(ref_torsion_edge_index=npy(te_ref[..., 1, :]), ref_sc_torsion_edge_mask=npy(cm_ref))
    # ligand graph: directed bonds sorted by src * N + dst (ligand.py:568-570), torsions by the reference's find_torsion
    directed = sorted([(a, b) for a, b in bonds] + [(b, a) for a, b in bonds], key=lambda e: e[0] * n_l + e[1])
    ei = np.asarray(directed, np.int64).T
    fake = types.SimpleNamespace(bond_prop=dict(bond_label=np.zeros(ei.shape[1], int)), edge_index=ei, numatoms=n_l,
                                 atomtype=np.zeros(n_l, int))
    tor_ref, rot_ref = du.find_torsion(fake)
    for what, fn in (("oracle", olig.find_torsion), ("product", plig.torsion_masks)):
        tor, rot = fn(n_l, ei)
        close(tor.astype(float), tor_ref.astype(float), 0.0, f"3dbs/find_torsion tor_edge_mask ({what})")
        close(rot.astype(float), rot_ref.astype(float), 0.0, f"3dbs/find_torsion rot_node_mask ({what})")
    rng = np.random.default_rng(5)
    for i in range(6):                                  # synthetic molecules incl. equal-halves ties
        lg = synthetic.make_ligand(rng, int(rng.integers(4, 40)))
        e2 = lg["lig_edge_index"]
        fk = types.SimpleNamespace(bond_prop=dict(bond_label=np.zeros(e2.shape[1], int)), edge_index=e2, numatoms=lg["n_lig"],
                                   atomtype=np.zeros(lg["n_lig"], int))
        tr, rr = du.find_torsion(fk)
        for fn in (olig.find_torsion, plig.torsion_masks):
            t2, r2 = fn(lg["n_lig"], e2)
            assert np.array_equal(t2, tr) and np.array_equal(r2, rr), ("find_torsion", i, fn.__module__)
    print("  pinned find_torsion on 6 synthetic molecules (oracle and product)")
    out.update(lig_edge_index=ei, ref_tor_edge_mask=tor_ref, ref_rot_node_mask=rot_ref)
    # edge counts at the crystal pose (SURVEY.md section 8 quotes 9 002 pocket edges and 458 ligand radius edges)
    rec = torch.from_numpy(pos[msk]).float()
    e_aa = cluster.radius_graph(rec, 4.0, torch.zeros(rec.shape[0], dtype=torch.long), max_num_neighbors=1000).shape[1]
    e_ll = cluster.radius_graph(torch.from_numpy(lig).float(), 5.0, torch.zeros(n_l, dtype=torch.long), max_num_neighbors=32).shape[1]
    print(f"  edges at the crystal pose: pocket {e_aa}, ligand radius {e_ll}, ligand bonds {ei.shape[1]}")
    out.update(e_aa=np.asarray(e_aa), e_ll=np.asarray(e_ll))
    np.savez_compressed(os.path.join(HERE, "real_3dbs.npz"), **out)



def golden_real_trajectory():
    """BASELINE config 1 end to end: the 3DBS pocket (866 atoms, reference-built templates / edges / features,
    Decentration) + its crystal ligand, 2 poses initialised by the (pinned) LigInit / SCProtInit restatement, taken
    through the reference's OWN `DiffBindFR.sample()` (scFlex.py:124-250 on tpscore.py) for all 20 steps.  Asserts the
    oracle reproduces it and freezes batch + noise + trajectories as tests/golden/real_3dbs_traj.npz."""
    print("[real complex 3DBS: 20-step reference trajectory]")
    import time
    from oracle import pose_init as opi
    sys.path.insert(0, os.path.join(HERE, ".."))
    z = np.load(os.path.join(HERE, "real_3dbs.npz"))
    if "druglib.datasets.builder" not in sys.modules or not hasattr(sys.modules["druglib.datasets.builder"], "PIPELINES"):
        dsb = types.ModuleType("druglib.datasets.builder")
        dsb.PIPELINES = ns.builder.INTERACTION.__class__("pipeline")
        sys.modules["druglib.datasets.builder"] = dsb
    obj = sys.modules["druglib.utils.obj"]
    obj.Ligand3D = object
    obj.make_torsion_mask = ns.prot_math.make_torsion_mask
    ref_shims._load("druglib.datasets.Docking.utils", "datasets/Docking/utils.py")
    pp = ref_shims._load("druglib.datasets.Docking.pocket_pipeline", "datasets/Docking/pocket_pipeline.py")
    seq, pos, msk = z["aatype"], z["atom14_position"], z["atom14_mask"]
    N = len(seq)
    seq_t, msk_t = torch.from_numpy(seq), torch.from_numpy(msk)
    fake = types.SimpleNamespace(num_res=lambda: N, atom_mask=np.zeros((N, 37)), residue_prop={})
    feat = pp.PocketFeaturizer()(dict(atom14_mask=msk_t, sequence=seq_t, pocket=fake))["pocket_node_feature"]
    data = dict(atom14_position=torch.from_numpy(pos).float(), atom14_mask=msk_t, sequence=seq_t,
                backbone_transl=torch.from_numpy(z["ref_backbone_transl"]).float(), lig_pos=torch.from_numpy(z["lig_pos"]).float())
    data = pp.Decentration()(data)
    import tests.test_real_complex as trc
    lig_half = trc._ligand_half(z)
    rec = dict(lig_half, lig_pos=data["lig_pos"], sequence=seq_t, atom14_mask=msk_t,
               backbone_transl=data["backbone_transl"], backbone_rots=torch.from_numpy(z["ref_backbone_rots"]).float(),
               default_frame=torch.from_numpy(z["ref_default_frame"]).float(),
               rigid_group_positions=torch.from_numpy(z["ref_rigid_group_positions"]).float(),
               torsion_angle=torch.from_numpy(z["ref_torsion_angle"]).float(),
               torsion_edge_index=torch.from_numpy(z["ref_torsion_edge_index"]),
               sc_torsion_edge_mask=torch.from_numpy(z["ref_sc_torsion_edge_mask"]), pocket_node_feature=feat)
    Tt = {k: (torch.from_numpy(np.asarray(v)) if k == "atom14_to_group" else v) for k, v in T.items()}
    fixed = opi.sc_fixer(copy.deepcopy(rec), T)
    rng = np.random.default_rng(31)
    n_tor = int(rec["tor_edge_mask"].sum())
    poses = []
    for _ in range(2):
        tape = dict(tor=rng.uniform(-np.pi, np.pi, n_tor), rot=synthetic._rand_rot(rng),
                    tr=torch.from_numpy(rng.normal(0, 3.0, (1, 3))).float(),           # near the pocket: cross edges on every step
                    sc=rng.uniform(-np.pi, np.pi, (N, 4)))
        poses.append(opi.init_pose(fixed, tape, Tt))
    coll = opi.collate(poses)
    d = types.SimpleNamespace(**coll)
    d.batch = d.lig_node_batch
    G = 2
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=1)
    model = ns.tpscore.TensorProductModel(ref_model_cfg()).eval()
    model.load_state_dict(params, strict=True)
    scfg = schedule.default_sample_cfg()
    go = sys.modules["druglib.utils.geometry_utils"]
    go.so3 = types.SimpleNamespace(score_norm=schedule.so3_score_norm)
    go.torus = types.SimpleNamespace(score_norm=lambda s: schedule.torus_score_norm(s, 0))
    ref_sampler = ns.scflex.DiffBindFR(diffusion_model=None, test_cfg=ED(sample_cfg=ED(vars(scfg))))
    ref_sampler.diffusion_model_cfg = ED(no_sc_torsion=False)
    ref_sampler.diffusion_model = model
    rd = ED({k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(d).items() if k != "rot_node_mask"})
    rd.metastore = {"rot_node_mask": [m.clone() for m in d.rot_node_mask]}
    torch.manual_seed(1234)
    t0 = time.time()
    res = ref_sampler.sample(rd, visualize=True)
    print(f"  reference sample(): 2 poses x {scfg.actual_steps} steps of the 866-atom pocket in {time.time() - t0:.0f}s")
    lig_ref = torch.cat([r[0] for r in res], dim=1)
    a14_ref = torch.cat([r[1] for r in res], dim=1)
    n_sc = int(d.sc_torsion_edge_mask.sum())
    noise = sampler.draw_noise(scfg.actual_steps, G, G * n_tor, n_sc, seed=1234)
    lig_o, a14_o = sampler.sample(params, mcfg, scfg, copy.deepcopy(d), noise,
                                  torch.from_numpy(T["atom14_to_group"]).long(), torus_seed=0, visualize=True)
    close(lig_o, lig_ref, 1e-4, "3dbs sample(): ligand trajectory (20 steps)")
    close(a14_o, a14_ref, 1e-4, "3dbs sample(): atom14 trajectory (20 steps)")
    out = batch_to_npz(d)
    out.update(params_seed=np.asarray(1), params_sha256=np.asarray(params_digest(params)), noise_tr=npy(noise.tr),
               noise_rot=npy(noise.rot), noise_tor=npy(noise.tor), noise_sc=npy(noise.sc), traj_lig=npy(lig_ref),
               final_atom14=npy(a14_ref[-1]), atom14_step0=npy(a14_ref[0]))
    for k in ("default_frame", "rigid_group_positions"):      # float32 exactly; compressed well
        out[k] = out[k].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "real_3dbs_traj.npz"), **out)
    print("  real_3dbs_traj.npz:", os.path.getsize(os.path.join(HERE, "real_3dbs_traj.npz")) // 1024, "KiB")


# --------------------------------------------------------------------------- 8b. the reference's own multi-complex examples
def _protein_superset(pdb, ref_xyz, pocket_cut=12.0, keep_cut=16.0):
    """One example receptor: (seq, atom14 pos, atom14 mask) of its 12 A pocket exactly as golden_real_complex selects it (any heavy atom
    within 12 A of any heavy atom of the pocket-defining ligand), and the atom37 arrays of every residue within `keep_cut` A -- the
    input the PRODUCT's own selection (pocket.pockets_from_proteins) starts from, a superset of the pocket."""
    pc = ns.pc
    seq, pos, msk, aa37, p37, m37 = [], [], [], [], [], []
    for key, rn, atoms in _parse_pdb_residues(pdb):
        if rn not in pc.restype_3to1:
            continue
        P = np.array(list(atoms.values()))
        dmin = np.linalg.norm(P[:, None] - ref_xyz[None], axis=-1).min()
        if dmin >= keep_cut:
            continue
        aa = pc.restype_order[pc.restype_3to1[rn]]
        row, mrow = np.zeros((37, 3), np.float32), np.zeros(37, bool)
        for nm, xyz in atoms.items():
            if nm in pc.atom_order:
                row[pc.atom_order[nm]] = xyz
                mrow[pc.atom_order[nm]] = True
        aa37.append(aa); p37.append(row); m37.append(mrow)
        if dmin >= pocket_cut:
            continue
        names = pc.restype_name_to_atom14_names[rn]
        seq.append(aa)
        pos.append([atoms.get(nm, [0.0, 0.0, 0.0]) if nm else [0.0, 0.0, 0.0] for nm in names])
        msk.append([bool(nm) and nm in atoms for nm in names])
    return (np.asarray(seq), np.asarray(pos, np.float64), np.asarray(msk)), (np.asarray(aa37), np.asarray(p37), np.asarray(m37))


def _ref_pocket_half(seq, pos, msk, pp, du):
    """Pocket half of a per-complex record through the reference's own functions (as golden_real_trajectory): templates, chi
    masks / edges, PocketFeaturizer, Decentration.  Returns (record dict, pocket centre)."""
    N = len(seq)
    ideal = T["atom14_mask"][seq][..., None].astype(np.float32)
    tpl = ns.prot_math.extract_chi_and_template(seq, pos.copy(), ideal, return_radian=True)
    seq_t, msk_t = torch.from_numpy(seq), torch.from_numpy(msk)
    te, cm = du.build_torsion_edges(seq_t, msk_t)
    fake = types.SimpleNamespace(num_res=lambda: N, atom_mask=np.zeros((N, 37)), residue_prop={})
    feat = pp.PocketFeaturizer()(dict(atom14_mask=msk_t, sequence=seq_t, pocket=fake))["pocket_node_feature"]
    data = dict(atom14_position=torch.from_numpy(pos).float(), atom14_mask=msk_t, sequence=seq_t,
                backbone_transl=torch.from_numpy(npy(tpl["backbone_transl"])).float(), lig_pos=torch.zeros(1, 3))
    data = pp.Decentration()(data)
    return dict(sequence=seq_t, atom14_mask=msk_t, backbone_transl=data["backbone_transl"],
                backbone_rots=torch.from_numpy(npy(tpl["backbone_rots"])).float(),
                default_frame=torch.from_numpy(npy(tpl["default_frame"])).float(),
                rigid_group_positions=torch.from_numpy(npy(tpl["rigid_group_positions"])).float(),
                torsion_angle=torch.from_numpy(npy(tpl["torsion_angle"])).float(),
                torsion_edge_index=te[..., 1, :], sc_torsion_edge_mask=cm, pocket_node_feature=feat)


def _ligand_half_from_sdf(path, du, seed):
    """Real geometry / bond graph / torsions (the reference's find_torsion) of one example ligand; seeded synthetic node and
    bond features (the RDKit featuriser is outside this path, as in tests/test_real_complex.py)."""
    xyz, bonds = _parse_sdf_heavy(path)
    n_l = xyz.shape[0]
    directed = sorted([(a, b) for a, b in bonds] + [(b, a) for a, b in bonds], key=lambda e: e[0] * n_l + e[1])
    ei = np.asarray(directed, np.int64).T
    fake = types.SimpleNamespace(bond_prop=dict(bond_label=np.zeros(ei.shape[1], int)), edge_index=ei, numatoms=n_l, atomtype=np.zeros(n_l, int))
    tor, rot = du.find_torsion(fake)
    rng = np.random.default_rng(seed)
    node = np.clip(rng.standard_normal((n_l, 27)), -3, 3).astype(np.float32)
    und, feat = {}, np.zeros((ei.shape[1], 10), np.float32)
    for k, (u, v) in enumerate(ei.T.tolist()):
        key = (min(u, v), max(u, v))
        if key not in und:
            und[key] = (int(rng.integers(0, 4)), rng.integers(0, 2, 4))
        feat[k, und[key][0]] = 1.0
        feat[k, 6:] = und[key][1]
    return dict(lig_node=torch.from_numpy(node), lig_pos=torch.from_numpy(xyz).float(), lig_edge_index=torch.from_numpy(ei),
                lig_edge_feat=torch.from_numpy(feat), tor_edge_mask=torch.from_numpy(np.asarray(tor)).long(),
                rot_node_mask=torch.from_numpy(np.asarray(rot)))


class _EdgeRecorder:
    """Wraps the graph builders of the REFERENCE's TensorProductModel instance (tpscore.py:575-759) for one sample() run and logs, per
    step and per graph, (1) the edge count of each of the five per-step edge sets -- lig = bonds + radius_graph(5 A, cap 32), atom =
    radius_graph(4 A), cross = lig x {CA, CB} + radius(0.2 sigma + 5 A), tor / sc = bond mid-point -> atoms within 5 / 4 A (cap 32) --
    and (2) the set's MARGIN: the smallest | |x_i - x_j| - cutoff | over all candidate pairs of the graph, evaluated in float64 on the
    coordinates the reference holds at that step.  A margin below ~1e-4 A means a pair sits within rounding distance of a hard cutoff:
    an implementation whose coordinates differ in the 5th decimal may legitimately build a different graph at that step
    (tests/test_examples.py).  The margins stand in for the per-step coordinates themselves (3 MB per fixture)."""
    SETS = ("lig", "atom", "cross", "tor", "sc")

    def __init__(self, model, G):
        self.m, self.G, self.counts, self.margins = model, G, [], []
        self._orig = {}

    def _per_graph(self, batch, idx):
        return np.bincount(npy(batch[idx]).astype(np.int64), minlength=self.G)[:self.G]

    def _margin(self, xa, ba, xb, bb, cut, exclude_self=False):
        out = np.full(self.G, np.inf)
        xa, xb, ba, bb = npy(xa).astype(np.float64), npy(xb).astype(np.float64), npy(ba), npy(bb)
        for g in range(self.G):
            A, B = xa[ba == g], xb[bb == g]
            if len(A) == 0 or len(B) == 0:
                continue
            d = np.sqrt(((A[:, None, :] - B[None, :, :]) ** 2).sum(-1))
            if exclude_self:
                d = d[~np.eye(len(A), dtype=bool)]
            c = cut[g] if np.ndim(cut) else cut
            if d.size:
                out[g] = np.abs(d - c).min()
        return out

    def install(self):
        m, R = self.m, self

        def lig(data):
            r = R._orig["build_lig_conv_graph"](data)
            R.counts.append({}); R.margins.append({})
            R.counts[-1]["lig"] = R._per_graph(data.lig_node_batch, r[1][0])
            R.margins[-1]["lig"] = R._margin(data.lig_pos, data.lig_node_batch, data.lig_pos, data.lig_node_batch, float(m.lig_cutoff), True)
            return r

        def atom(data):
            r = R._orig["build_atom_conv_graph"](data)
            R.counts[-1]["atom"] = R._per_graph(data.rec_atm_pos_batch, r[1][0])
            R.margins[-1]["atom"] = R._margin(data.rec_atm_pos, data.rec_atm_pos_batch, data.rec_atm_pos, data.rec_atm_pos_batch, float(m.atom_cutoff), True)
            return r

        def cross(data):
            r = R._orig["build_cross_conv_graph"](data)
            R.counts[-1]["cross"] = R._per_graph(data.lig_node_batch, r[0][0])
            a37 = data.pocket_node_feature[:, 0].long()
            nab = ~((a37 == 1) | (a37 == 3))
            cut = npy(data.tr_sigma).astype(np.float64) * np.float64(np.float32(0.2)) + 5.0 if m.dynamic_max_cross else float(m.cross_cutoff)
            R.margins[-1]["cross"] = R._margin(data.lig_pos, data.lig_node_batch, data.rec_atm_pos[nab], data.rec_atm_pos_batch[nab], cut)
            return r

        def tor(data, x):
            r = R._orig["build_lig_bond_conv_graph"](data, x)
            bonds = data.lig_edge_index[:, data.tor_edge_mask.bool()]
            bb = data.lig_node_batch[bonds[0]]
            R.counts[-1]["tor"] = R._per_graph(bb, r[0][0])
            mid = (data.lig_pos[bonds[0]] + data.lig_pos[bonds[1]]) / 2
            R.margins[-1]["tor"] = R._margin(mid, bb, data.lig_pos, data.lig_node_batch, float(m.lig_cutoff))
            return r

        def sc(data, x):
            r = R._orig["build_sc_bond_conv_graph"](data, x)
            bonds = data.sc_torsion_edge_index
            bb = data.rec_atm_pos_batch[bonds[0]]
            R.counts[-1]["sc"] = R._per_graph(bb, r[0][0])
            mid = (data.rec_atm_pos[bonds[0]] + data.rec_atm_pos[bonds[1]]) / 2
            R.margins[-1]["sc"] = R._margin(mid, bb, data.rec_atm_pos, data.rec_atm_pos_batch, float(m.atom_cutoff))
            return r

        for name, fn in (("build_lig_conv_graph", lig), ("build_atom_conv_graph", atom), ("build_cross_conv_graph", cross),
                         ("build_lig_bond_conv_graph", tor), ("build_sc_bond_conv_graph", sc)):
            self._orig[name] = getattr(m, name)
            object.__setattr__(m, name, fn)
        return self

    def arrays(self):
        """(edge_counts [steps, G, 5] int32, cutoff_margin [steps, G, 5] float32 (inf: the graph has no candidate pair in that set))"""
        S = len(self.counts)
        c = np.zeros((S, self.G, 5), np.int32)
        mg = np.full((S, self.G, 5), np.inf, np.float32)
        for s in range(S):
            for k, nm in enumerate(self.SETS):
                if nm in self.counts[s]:
                    c[s, :, k] = self.counts[s][nm]
                    mg[s, :, k] = self.margins[s][nm]
        return c, mg


def _reference_trajectories(recs, seed, what, run_oracle=True):
    """One pose per record (seeded init tape through the pinned LigInit / SCProtInit restatement), all poses in ONE batch through the
    reference's own DiffBindFR.sample() for 20 steps (its graph builders wrapped by _EdgeRecorder); the oracle must reproduce it.
    Returns (tapes, noise, traj_lig, final_atom14, params, (edge_counts, cutoff_margin))."""
    import time
    from oracle import pose_init as opi
    Tt = {k: (torch.from_numpy(np.asarray(v)) if k == "atom14_to_group" else v) for k, v in T.items()}
    rng = np.random.default_rng(seed)
    poses, tapes = [], []
    for rec in recs:
        fixed = opi.sc_fixer(copy.deepcopy(rec), T)
        n_tor, N = int(rec["tor_edge_mask"].sum()), int(rec["sequence"].shape[0])
        tape = dict(tor=rng.uniform(-np.pi, np.pi, n_tor), rot=synthetic._rand_rot(rng),
                    tr=torch.from_numpy(rng.normal(0, 3.0, (1, 3))).float(), sc=rng.uniform(-np.pi, np.pi, (N, 4)))
        tapes.append(tape)
        poses.append(opi.init_pose(fixed, tape, Tt))
    d = types.SimpleNamespace(**opi.collate(poses))
    d.batch = d.lig_node_batch
    G = len(recs)
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=1)
    model = ns.tpscore.TensorProductModel(ref_model_cfg()).eval()
    model.load_state_dict(params, strict=True)
    scfg = schedule.default_sample_cfg()
    go = sys.modules["druglib.utils.geometry_utils"]
    go.so3 = types.SimpleNamespace(score_norm=schedule.so3_score_norm)
    go.torus = types.SimpleNamespace(score_norm=lambda s: schedule.torus_score_norm(s, 0))
    ref_sampler = ns.scflex.DiffBindFR(diffusion_model=None, test_cfg=ED(sample_cfg=ED(vars(scfg))))
    ref_sampler.diffusion_model_cfg = ED(no_sc_torsion=False)
    ref_sampler.diffusion_model = model
    rd = ED({k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(d).items() if k != "rot_node_mask"})
    rd.metastore = {"rot_node_mask": [m.clone() for m in d.rot_node_mask]}
    torch.manual_seed(seed)
    t0 = time.time()
    recorder = _EdgeRecorder(model, G).install()
    res = ref_sampler.sample(rd, visualize=True)
    print(f"  reference sample(): {G} poses x {scfg.actual_steps} steps ({what}) in {time.time() - t0:.0f}s")
    edges = recorder.arrays()
    assert edges[0].shape[0] == scfg.actual_steps
    lig_ref = torch.cat([r[0] for r in res], dim=1)
    a14_ref = torch.cat([r[1] for r in res], dim=1)
    n_tor_tot, n_sc = int(d.tor_edge_mask.sum()), int(d.sc_torsion_edge_mask.sum())
    noise = sampler.draw_noise(scfg.actual_steps, G, n_tor_tot, n_sc, seed=seed)
    if not run_oracle:
        return tapes, noise, lig_ref, a14_ref[-1], params, edges
    lig_o, a14_o = sampler.sample(params, mcfg, scfg, copy.deepcopy(d), noise, torch.from_numpy(T["atom14_to_group"]).long(),
                                  torus_seed=0, visualize=True)
    close(lig_o, lig_ref, 1e-4, f"{what} sample(): ligand trajectories (20 steps)")
    close(a14_o, a14_ref, 1e-4, f"{what} sample(): atom14 trajectories (20 steps)")
    return tapes, noise, lig_ref, a14_ref[-1], params, edges


def golden_examples():
    """The reference's own multi-complex examples (README.md:86-127) as fixtures: examples/forward = the 3DBS receptor with all 15
    SDF ligands (BASELINE config 3 in miniature: one shared pocket), examples/reverse = 2 ligands x 3 receptors (config 4: shared
    ligands), one pose each, all 20 steps through the reference's own sample().  The fixtures hold the RAW inputs of the product
    path (protein atom37 arrays around the site, the pocket-defining crystal ligand, ligand graphs), the tapes and the reference
    trajectories: tests/golden/real_forward15_traj.npz, real_reverse_traj.npz."""
    print("[examples/forward and examples/reverse: 20-step reference trajectories]")
    sys.path.insert(0, os.path.join(HERE, ".."))
    if "druglib.datasets.builder" not in sys.modules or not hasattr(sys.modules["druglib.datasets.builder"], "PIPELINES"):
        dsb = types.ModuleType("druglib.datasets.builder")
        dsb.PIPELINES = ns.builder.INTERACTION.__class__("pipeline")
        sys.modules["druglib.datasets.builder"] = dsb
    obj = sys.modules["druglib.utils.obj"]
    obj.Ligand3D = object
    obj.make_torsion_mask = ns.prot_math.make_torsion_mask
    du = ref_shims._load("druglib.datasets.Docking.utils", "datasets/Docking/utils.py")
    pp = ref_shims._load("druglib.datasets.Docking.pocket_pipeline", "datasets/Docking/pocket_pipeline.py")
    ex = os.path.join(ref_shims.REF, "examples")

    HALF_KEYS = ("sequence", "atom14_mask", "backbone_transl", "backbone_rots", "default_frame", "rigid_group_positions", "torsion_angle",
                 "torsion_edge_index", "sc_torsion_edge_mask", "pocket_node_feature")

    def halves_of(prots_pk):
        """prot{i}_half_<key>: the pocket half the REFERENCE's pipeline built (what the frozen trajectories started from)."""
        return {f"prot{i}_half_{k}": npy(pk[k]) for i, pk in enumerate(prots_pk) for k in HALF_KEYS}

    if os.environ.get("GOLDEN_EXAMPLES_HALVES_ONLY"):      # add the reference pocket halves to existing fixtures without re-running sample()
        for name, sub, names in (("real_forward15_traj.npz", "forward", ["3dbs"]), ("real_reverse_traj.npz", os.path.join("reverse", "receptors"), ["2src", "3mhw", "3pp0"])):
            z = dict(np.load(os.path.join(HERE, name)))
            pks = []
            for nm in names:
                c, _ = _parse_sdf_heavy(os.path.join(ex, sub, f"{nm}_protein_crystal.sdf"))
                (seq, pos, msk), _sup = _protein_superset(os.path.join(ex, sub, f"{nm}_protein.pdb"), c)
                pks.append(_ref_pocket_half(seq, pos, msk, pp, du))
            z.update(halves_of(pks))
            np.savez_compressed(os.path.join(HERE, name), **z)
            print(f"  {name}: reference pocket halves added, {os.path.getsize(os.path.join(HERE, name)) // 1024} KiB")
        return

    EDGES_ONLY = bool(os.environ.get("GOLDEN_EXAMPLES_EDGES_ONLY"))   # add edge_counts / cutoff_margin to the existing fixtures: the
    # reference's sample() is run again (recorder installed) and must reproduce the frozen trajectories; the oracle leg is skipped

    def save(name, prots, ligs, pairs, tapes, noise, lig_traj, a14_final, params, prots_pk, edges):
        if EDGES_ONLY:
            z = dict(np.load(os.path.join(HERE, name)))
            dl = float(np.abs(z["traj_lig"] - npy(lig_traj).astype(np.float32)).max())
            da = float(np.abs(z["final_atom14"] - npy(a14_final).astype(np.float32)).max())
            print(f"  {name}: re-run vs frozen trajectories max|d| ligand {dl:.2e} A, atom14 {da:.2e} A")
            # (fp32 BLAS summation order moves with the thread count / build: 3e-5 A between two runs of the reference itself)
            assert dl <= 1e-4 and da <= 1e-4, "the re-run does not reproduce the frozen reference trajectories"
            # one consistent reference run: the trajectories of THIS run go into the fixture together with its edge counts
            z.update(edge_counts=edges[0], cutoff_margin=edges[1], traj_lig=npy(lig_traj).astype(np.float32),
                     final_atom14=npy(a14_final).astype(np.float32))
            np.savez_compressed(os.path.join(HERE, name), **z)
            print(f"  {name}: edge_counts {edges[0].shape} / cutoff_margin added, {os.path.getsize(os.path.join(HERE, name)) // 1024} KiB")
            return
        out = dict(edge_counts=edges[0], cutoff_margin=edges[1], n_prot=np.asarray(len(prots)), n_lig=np.asarray(len(ligs)), pairs=np.asarray(pairs, np.int64),
                   params_seed=np.asarray(1), params_sha256=np.asarray(params_digest(params)),
                   noise_tr=npy(noise.tr), noise_rot=npy(noise.rot), noise_tor=npy(noise.tor), noise_sc=npy(noise.sc),
                   traj_lig=npy(lig_traj).astype(np.float32), final_atom14=npy(a14_final).astype(np.float32))
        out.update(halves_of(prots_pk))
        for i, (nm, cry, sup) in enumerate(prots):
            out.update({f"prot{i}_name": np.asarray(nm), f"prot{i}_ref_lig_pos": cry.astype(np.float32), f"prot{i}_aatype": sup[0],
                        f"prot{i}_atom37_pos": sup[1].astype(np.float32), f"prot{i}_atom37_mask": sup[2]})
        for i, (nm, lg) in enumerate(ligs):
            out[f"lig{i}_name"] = np.asarray(nm)
            out.update({f"lig{i}_{k}": npy(v) for k, v in lg.items()})
        for g, t in enumerate(tapes):
            out.update({f"tape{g}_tor": np.asarray(t["tor"]), f"tape{g}_rot": np.asarray(t["rot"]), f"tape{g}_tr": npy(t["tr"]),
                        f"tape{g}_sc": np.asarray(t["sc"])})
        path = os.path.join(HERE, name)
        np.savez_compressed(path, **out)
        print(f"  {name}: {os.path.getsize(path) // 1024} KiB, {len(pairs)} pairs")

    # ---- forward: one receptor, 15 ligands
    cry, _ = _parse_sdf_heavy(os.path.join(ex, "forward", "3dbs_protein_crystal.sdf"))
    (seq, pos, msk), sup = _protein_superset(os.path.join(ex, "forward", "3dbs_protein.pdb"), cry)
    assert len(seq) == 105 and int(msk.sum()) == 866
    pk = _ref_pocket_half(seq, pos, msk, pp, du)
    files = sorted(os.listdir(os.path.join(ex, "forward", "mols")))
    assert len(files) == 15
    ligs = [(f[:-4], _ligand_half_from_sdf(os.path.join(ex, "forward", "mols", f), du, 100 + i)) for i, f in enumerate(files)]
    recs = [dict(pk, **lg) for _, lg in ligs]
    tapes, noise, lt, a14, params, edges = _reference_trajectories(recs, 4321, "examples/forward (3DBS x 15 ligands)", run_oracle=not EDGES_ONLY)
    save("real_forward15_traj.npz", [("3dbs", cry, sup)], ligs, [(0, i) for i in range(15)], tapes, noise, lt, a14, params, [pk], edges)

    # ---- reverse: 2 ligands x 3 receptors (pairs ligand-major, as dataframe.py builds the table)
    prots, pks = [], []
    for nm in ("2src", "3mhw", "3pp0"):
        c, _ = _parse_sdf_heavy(os.path.join(ex, "reverse", "receptors", f"{nm}_protein_crystal.sdf"))
        (seq, pos, msk), sup = _protein_superset(os.path.join(ex, "reverse", "receptors", f"{nm}_protein.pdb"), c)
        print(f"  {nm}: pocket {len(seq)} residues / {int(msk.sum())} atoms, superset {len(sup[0])} residues")
        prots.append((nm, c, sup))
        pks.append(_ref_pocket_half(seq, pos, msk, pp, du))
    ligs = [(f"ligand_{i + 1}", _ligand_half_from_sdf(os.path.join(ex, "reverse", f"ligand_{i + 1}.sdf"), du, 200 + i)) for i in range(2)]
    pairs = [(p, l) for l in range(2) for p in range(3)]
    recs = [dict(pks[p], **ligs[l][1]) for p, l in pairs]
    tapes, noise, lt, a14, params, edges = _reference_trajectories(recs, 8765, "examples/reverse (2 ligands x 3 receptors)", run_oracle=not EDGES_ONLY)
    save("real_reverse_traj.npz", prots, ligs, pairs, tapes, noise, lt, a14, params, pks, edges)



# --------------------------------------------------------------------------- 8. output side
def _parse_sdf_elements(path):
    L = open(path).read().split("\n")
    na = int(L[3][0:3])
    return [L[4 + i][31:34].strip() for i in range(na) if L[4 + i][31:34].strip() != "H"]


def _load_ref_export():
    """The reference's own output-side functions: druglib/utils/obj/protein.py (Protein, to_pdb) and the leaf files of
    DiffBindFR/metrics that import without rdkit / openfold (centroid, scrmsd, lrmsd.symm_rmsd)."""
    import contextlib
    import importlib.util
    for nm in ["Bio.PDB.ResidueDepth", "Bio.PDB.DSSP", "Bio.PDB.Model", "druglib.ops", "druglib.ops.dssp",
               "druglib.ops.msms", "druglib.alerts", "rdkit.Chem.rdFMCS"]:
        if nm not in sys.modules:
            sys.modules[nm] = ref_shims._Anything(nm)
    import druglib
    druglib.__version__ = "1.0.0"
    druglib.time_limit = lambda seconds: contextlib.nullcontext()
    prot = ref_shims._load("druglib.utils.obj.protein", "utils/obj/protein.py")
    mods = {}
    for leaf in ("centroid", "scrmsd", "lrmsd"):
        spec = importlib.util.spec_from_file_location("ref_metrics_" + leaf,
                                                      os.path.join(ref_shims.COPY, "DiffBindFR", "metrics", leaf + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[leaf] = m
    src = open(os.path.join(ref_shims.REF, "DiffBindFR", "common", "inference_dataset.py")).read()
    a = src.index("def add_center_pos(")
    env = {"Tensor": torch.Tensor}
    exec(compile(src[a: src.index("# EValuation and Inference Data Container")], "inference_dataset.py", "exec"), env)
    return prot, mods, env["add_center_pos"]


def golden_export():
    """Row f3: per-pose metrics + PDB text of `complex_modeling` (DiffBindFR/evaluation/export.py:106-312) on the 3DBS
    example: protein = the residues within 20 A of the crystal ligand (two artificial chains), pocket = within 12 A."""
    print("[export 3DBS]")
    from oracle import export as oex
    prot, mods, add_center = _load_ref_export()
    pc = ns.pc
    ex = os.path.join(ref_shims.REF, "examples", "forward")
    lig, bonds = _parse_sdf_heavy(os.path.join(ex, "3dbs_protein_crystal.sdf"))
    elements = _parse_sdf_elements(os.path.join(ex, "3dbs_protein_crystal.sdf"))
    lig = lig.astype(np.float32)
    bf_of = {}
    for l in open(os.path.join(ex, "3dbs_protein.pdb")):
        if l.startswith("ATOM"):
            bf_of.setdefault((l[21], int(l[22:26]), l[26], l[12:16].strip()), float(l[60:66]))
    aatype, pos37, m37, resid, bfac, in_pocket = [], [], [], [], [], []
    for key, rn, atoms in _parse_pdb_residues(os.path.join(ex, "3dbs_protein.pdb")):
        if rn not in pc.restype_3to1:
            continue
        P = np.array(list(atoms.values()))
        dmin = np.linalg.norm(P[:, None] - lig[None], axis=-1).min()
        if dmin >= 20.0:
            continue
        aatype.append(pc.restype_order[pc.restype_3to1[rn]])
        pos37.append([atoms.get(a, [0.0, 0.0, 0.0]) for a in pc.atom_types])
        m37.append([float(a in atoms) for a in pc.atom_types])
        bfac.append([bf_of.get(key + (a,), 0.0) for a in pc.atom_types])
        resid.append(key[1])
        in_pocket.append(dmin < 12.0)
    aatype = np.asarray(aatype, np.int64)
    pos37 = np.asarray(pos37, np.float32)
    m37 = np.asarray(m37, np.float32)
    bfac = np.asarray(bfac, np.float64)
    resid = np.asarray(resid, np.int64)
    pocket_mask = np.asarray(in_pocket)
    n = aatype.shape[0]
    chain = (np.arange(n) >= n // 2).astype(np.int64) * 27             # chains 'A' and 'BA' (int_id_to_str_id)
    aatype[3] = 20                                                      # one unknown residue -> 'UNK'
    for r in (n - 1, int(np.nonzero(pocket_mask)[0][-1])):              # terminal OXT: atom37 slot 36 has no atom14 slot, so
        m37[r, 36] = 1.0                                                # pos_update writes it at slot 0's (N) coordinates
        pos37[r, 36] = pos37[r, 2] + np.float32(1.2)
    print(f"  protein {n} residues / {int(m37.sum())} atoms, pocket {int(pocket_mask.sum())} residues, ligand {lig.shape[0]}")
    remark = "REMARK   1 CREATED WITH MDLDruglib 1.0.0, 2024-11-01"
    P = prot.Protein(name="3dbs", atom_positions=pos37, aatype=aatype, atom_mask=m37, residue_index=resid, b_factors=bfac,
                     chain_index=chain, remark=remark, pocket_mask=pocket_mask)
    pocket = prot.Protein(name="3dbs_pkt", atom_positions=pos37[pocket_mask], aatype=aatype[pocket_mask],
                          atom_mask=m37[pocket_mask], residue_index=resid[pocket_mask], b_factors=bfac[pocket_mask],
                          chain_index=chain[pocket_mask], remark=remark)
    # ---- trajectories: pocket-centred like the sampler's output (Decentration, pocket_pipeline.py:276-300)
    tgt14, tmask = pocket.to_pos14(True)
    tmask = tmask[..., 0]
    center = torch.from_numpy(tgt14[tmask > 0].mean(0)).float()
    tgt14_c = torch.from_numpy(tgt14).float() - center * torch.from_numpy(tmask)[..., None].float()
    seq = torch.from_numpy(aatype[pocket_mask])
    rng = np.random.default_rng(31)
    n_pose, n_frame = 4, 3
    sig = np.array([0.0, 0.05, 0.4, 1.5], np.float32)[:, None, None, None, None] * np.array([1.0, 0.5, 0.25], np.float32)[None, :, None, None, None]
    prot_traj = tgt14_c[None, None] + torch.from_numpy(rng.standard_normal((n_pose, n_frame) + tgt14.shape).astype(np.float32) * sig)
    swap = torch.from_numpy(T["atom14_swap"]).long()[seq]
    prot_traj[1, 2] = torch.gather(tgt14_c, 1, swap[..., None].expand(-1, -1, 3))   # the other naming: sc-rmsd must be 0
    prot_traj = prot_traj * torch.from_numpy(tmask)[None, None, :, :, None].float()
    lig_c = torch.from_numpy(lig) - center
    lig_traj = lig_c[None, None] + torch.from_numpy(rng.standard_normal((n_pose, n_frame) + lig.shape).astype(np.float32) * sig[..., 0, :, :] * 2)
    out = dict(aatype=aatype, atom37_pos=pos37, atom37_mask=m37, residue_index=resid, chain_index=chain, b_factors=bfac,
               pocket_mask=pocket_mask, center=npy(center), prot_traj=npy(prot_traj), lig_traj=npy(lig_traj), lig_pos=lig,
               target_atom14=npy(tgt14_c), target_atom14_mask=tmask, remark=np.asarray(remark))
    # ---- metrics exactly as complex_modeling calls them (export.py:139-195)
    lt, pt = add_center(lig_traj, center), add_center(prot_traj, center)
    close(oex.add_center_pos(lig_traj, center), lt, 0.0, "add_center_pos")
    cen_ref = mods["centroid"].calc_lig_centroid(lt, torch.from_numpy(lig).float())
    close(oex.calc_lig_centroid(lt, torch.from_numpy(lig).float()), cen_ref, 0.0, "calc_lig_centroid")
    tm = torch.from_numpy(tmask).float()
    sc_ref = mods["scrmsd"].sidechain_rmsd(pt, add_center(tgt14_c, center), tm, seq)
    close(oex.sidechain_rmsd(pt, oex.add_center_pos(tgt14_c, center), tm, seq, T), sc_ref, 0.0, "sidechain_rmsd")
    assert float(sc_ref[1, 2]) < 1e-5 and float(sc_ref[0, 0]) < 1e-5, "swapped naming / identical pose"
    alt_ref = mods["scrmsd"].make_altern_atom14(tgt14_c, tm, seq)
    alt = oex.make_altern_atom14(tgt14_c, tm, seq, T)
    close(alt[0], alt_ref[0], 0.0, "make_altern_atom14 pos")
    close(alt[1], alt_ref[1], 0.0, "make_altern_atom14 mask")
    out.update(ref_centroid=npy(cen_ref), ref_sc_rmsd=npy(sc_ref))
    # ---- symmetry-corrected ligand RMSD: the reference's graph matching gives the automorphisms (lrmsd.py:287-335)
    el_ids = np.asarray([sorted(set(elements)).index(e) for e in elements], np.int64)
    n_l = lig.shape[0]
    ei = np.asarray(sorted([(a, b) for a, b in bonds] + [(b, a) for a, b in bonds], key=lambda e: e[0] * n_l + e[1]), np.int64).T
    nxg = ns.torch_utils.to_nx(el_ids, ei)
    perms_ref = ns.torch_utils.match_graphs(nxg, nxg, keep_self=True)
    ha = np.ones(n_l, bool)
    ha[5] = False                                                                 # one atom flagged as hydrogen-like
    rm_ref = mods["lrmsd"].symm_rmsd(nxg, ha, lig, npy(lt))
    rm = oex.symm_rmsd(perms_ref, ha, lig, npy(lt))
    close(rm, rm_ref, 0.0, "symm_rmsd")
    print(f"  ligand automorphisms: {len(perms_ref)}")
    out.update(lig_elements=el_ids, lig_edge_index=ei, ha_mask=ha, ref_perms=np.stack([p[0] for p in perms_ref]),
               ref_symm_rmsd=npy(rm_ref))
    # ---- PDB text: prot_final.pdb / pkt_final.pdb of export.py:261-274 for two poses
    texts = {}
    for pid in (0, 3):
        fp14, _ = P.to_pos14(True)
        fp14[pocket_mask] = npy(pt[pid, -1])
        ref_full = P.pos_update(fp14, None, False).to_pdb()
        ref_pkt = pocket.pos_update(npy(pt[pid, -1]), None, False).to_pdb()
        mine_full = oex.pose_pdb(aatype, pos37, m37, resid, chain, bfac, np.nonzero(pocket_mask)[0], npy(pt[pid, -1]), T, remark)
        mine_pkt = oex.to_pdb(aatype[pocket_mask], oex.pos14_to_pos37(aatype[pocket_mask], npy(pt[pid, -1]), m37[pocket_mask], T),
                              m37[pocket_mask], resid[pocket_mask], chain[pocket_mask], bfac[pocket_mask], T, remark)
        assert mine_full == ref_full, f"to_pdb (protein, pose {pid})"
        assert mine_pkt == ref_pkt, f"to_pdb (pocket, pose {pid})"
        texts[f"ref_pdb_full_{pid}"] = np.frombuffer(ref_full.encode(), np.uint8)
        texts[f"ref_pdb_pkt_{pid}"] = np.frombuffer(ref_pkt.encode(), np.uint8)
    print(f"  pinned to_pdb on 2 poses x (protein {len(ref_full)} B, pocket {len(ref_pkt)} B)")
    out.update(texts)
    # a small synthetic strip: >26 chains, negative / 4-digit residue numbers, serial numbers past 99999 are not
    # reachable at this size, unknown residues
    r2 = np.random.default_rng(7)
    n2 = 60
    aa2 = r2.integers(0, 21, n2)
    m2 = (T["atom37_mask"][aa2] * (r2.random((n2, 37)) > 0.1)).astype(np.float32)
    p2 = (r2.standard_normal((n2, 37, 3)) * np.array([5.0, 80.0, 900.0])).astype(np.float32)
    p2[0, 0] = [0.0005, -0.0005, 1234.5675]
    p2[1, 0] = [0.0625, -9999.9995, 0.125]
    ch2 = np.sort(r2.integers(0, 30, n2))
    ri2 = r2.integers(-50, 3000, n2)
    bf2 = r2.random((n2, 37)) * 120
    S = prot.Protein(name="strip", atom_positions=p2, aatype=aa2, atom_mask=m2, residue_index=ri2, b_factors=bf2, chain_index=ch2,
                     remark=None)
    ref2 = S.to_pdb().split("\n", 1)[1]                      # drop the dated REMARK line
    mine2 = oex.to_pdb(aa2, p2, m2, ri2, ch2, bf2, T, None)
    assert mine2 == ref2, "to_pdb (synthetic strip)"
    out.update(strip_aatype=aa2, strip_mask=m2, strip_pos=p2, strip_chain=ch2, strip_resid=ri2, strip_bfac=bf2,
               ref_pdb_strip=np.frombuffer(ref2.encode(), np.uint8))
    np.savez_compressed(os.path.join(HERE, "export.npz"), **out)


def golden_pocket_select():
    """The residue selection ahead of row f2 (druglib/utils/bio_utils/select_pocket.py, Protein.query_region): the reference's
    own torch functions on the 281-residue 3DBS cut of export.npz (inputs live there), several cutoffs / neighbour caps /
    atom selections."""
    print("[pocket selection 3DBS]")
    from oracle import pocket as opk
    import importlib.util
    spec = importlib.util.spec_from_file_location("druglib.utils.bio_utils.select_pocket",
                                                  os.path.join(ref_shims.COPY, "druglib", "utils", "bio_utils", "select_pocket.py"))
    sp = importlib.util.module_from_spec(spec)
    sp.__package__ = "druglib.utils.bio_utils"
    spec.loader.exec_module(sp)
    z = np.load(os.path.join(HERE, "export.npz"))
    pos, msk, lig = torch.from_numpy(z["atom37_pos"]), torch.from_numpy(z["atom37_mask"]), torch.from_numpy(z["lig_pos"])
    out = {}
    cases = [("any12", 12.0, None, None), ("any8", 8.0, None, None), ("any12_top40", 12.0, 40, None), ("any3", 3.0, None, None),
             ("far", 0.5, None, None), ("ca10", 10.0, None, (1,)), ("bb9_top25", 9.0, 25, (0, 1, 2, 4))]
    for name, cut, cap, atoms in cases:
        if atoms is None:
            ref = sp.select_bs_any(lig.clone(), pos.clone(), msk.clone(), cutoff=cut, max_neighbors=cap)
            mine, d2 = opk.select_bs(lig, pos, msk, cut, cap)
        else:
            ref = sp.select_bs_atoms(lig.clone(), pos.clone(), msk.clone(), atoms, cutoff=cut, max_neighbors=cap)
            mine, d2 = opk.select_bs(lig, pos[:, list(atoms)], msk[:, list(atoms)], cut, cap)
        assert torch.equal(ref, mine), name
        out["ref_" + name] = npy(ref)
        out["d2_" + name] = npy(d2)
        print(f"  pinned select_bs {name:12s} {int(ref.sum()):4d} residues")
    assert int(out["ref_any12"].sum()) == int(z["pocket_mask"].sum()) and int(out["ref_far"].sum()) == 1
    cen = (pos * msk[..., None]).sum(1) / msk.sum(1, keepdim=True)
    ref = sp.select_bs_centroid(lig.clone(), cen.clone(), msk.clone(), cutoff=9.0, max_neighbors=None)
    mine, d2 = opk.select_bs(lig, cen[:, None], msk.bool().any(-1, keepdim=True), 9.0, None)
    assert torch.equal(ref, mine), "centroid"
    out["ref_centroid9"], out["d2_centroid9"], out["centroids"] = npy(ref), npy(d2), npy(cen)
    np.savez_compressed(os.path.join(HERE, "pocket_select.npz"), **out)





def golden_chi_differ():
    """f3: `chi_differ` (DiffBindFR/metrics/angbin.py:48-103) calls `atom37_to_torsion_angles` of the openfold package the
    reference VENDORS (openfold/data/data_transforms.py:943-): it imports here once `openfold.config` (needs ml_collections,
    absent; only four placeholder names are read) and `tree` are stood in.  Runs the reference's chi_differ on the 3DBS poses
    of export.npz, asserts the oracle's restatement, freezes the result (tests/golden/chi_differ.npz)."""
    print("[chi_differ through the vendored openfold transforms]")
    import importlib.util
    from oracle import export as oex
    cfgm = types.ModuleType("openfold.config")
    cfgm.NUM_RES, cfgm.NUM_EXTRA_SEQ, cfgm.NUM_TEMPLATES, cfgm.NUM_MSA_SEQ = "num residues placeholder", "extra", "templates", "msa"
    root = os.path.join(ref_shims.COPY, "openfold")
    of = types.ModuleType("openfold"); of.__path__ = [root]
    sys.modules["openfold"] = of
    sys.modules["openfold.config"] = cfgm
    for sub in ("np", "utils", "data", "resources"):
        m = types.ModuleType("openfold." + sub); m.__path__ = [os.path.join(root, sub)]
        sys.modules["openfold." + sub] = m
    import importlib
    dt = importlib.import_module("openfold.data.data_transforms")
    assert hasattr(dt, "atom37_to_torsion_angles")
    tu = sys.modules["druglib.utils.torch_utils"]
    assert hasattr(tu, "batched_gather")
    sys.modules["druglib.utils.obj"].prot_math = ns.prot_math
    spec = importlib.util.spec_from_file_location("ref_metrics_angbin", os.path.join(ref_shims.COPY, "DiffBindFR", "metrics", "angbin.py"))
    ang = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ang)
    z = np.load(os.path.join(HERE, "export.npz"))
    seq = torch.from_numpy(z["aatype"][z["pocket_mask"]])
    center = torch.from_numpy(z["center"])
    pred = torch.from_numpy(z["prot_traj"]) + center
    tgt = torch.from_numpy(z["target_atom14"]) + center
    tmask = torch.from_numpy(z["target_atom14_mask"])
    ref_d, ref_m = ang.chi_differ(pred, tgt, tmask, seq)
    d, m = oex.chi_differ(pred, tgt, tmask, seq, T)
    close(m.float().expand_as(ref_m), ref_m.float(), 0.0, "chi_differ/torsion mask")
    # fp32 coordinates ~50 A from the origin: the reference's Rigid algebra and the restatement's direct projection round
    # differently (openfold's own docstring: "extremely sensitive to floating point imprecisions"); 1e-4 rad = 0.006 degrees
    close(d, ref_d, 1e-4, "chi_differ/delta chi (radians)")
    rate_ref = ((ref_d < 15 / 180 * torch.pi) * ref_m.bool()).sum(dim=-2) / ref_m.sum(dim=-2)
    close(oex.chi_success_rate(d, m), rate_ref, 1e-7, "chi_differ/chi success rates (export.py:176-179)")
    # a second case: random rigid-group rotations about chi on the synthetic strip of all residue types
    np.savez_compressed(os.path.join(HERE, "chi_differ.npz"), ref_delta_chi=npy(ref_d), ref_mask=npy(ref_m[0, 0]), ref_chi_rate=npy(rate_ref))


# --------------------------------------------------------------------------- f4: MDN pose scorer (KarmaDock) network forward
def _pyg_standins():
    """torch_geometric / torch_scatter are absent offline: the few entry points the scorer's architecture files import,
    restated from their documented semantics (parity UNPINNED at this boundary, like e3nn)."""
    from oracle import cluster
    import torch.nn as nn

    class MessagePassing(nn.Module):
        """flow='source_to_target': x_j = x[edge_index[0]], x_i = x[edge_index[1]], aggregation over i."""
        def __init__(self, aggr="add", **kw):
            super().__init__()
            self.aggr = aggr

        def propagate(self, edge_index, **kw):
            import inspect
            j, i = edge_index[0], edge_index[1]
            args = {}
            for name in inspect.signature(self.message).parameters:
                if name.endswith("_i"):
                    args[name] = kw[name[:-2]][i]
                elif name.endswith("_j"):
                    args[name] = kw[name[:-2]][j]
                else:
                    args[name] = kw[name]
            msg = self.message(**args)
            n = next(v for k, v in kw.items() if torch.is_tensor(v) and k != "edge_attr").shape[0]
            out = torch.zeros((n,) + tuple(msg.shape[1:]), dtype=msg.dtype).index_add_(0, i, msg)
            if self.aggr == "mean":
                cnt = torch.zeros(n, dtype=msg.dtype).index_add_(0, i, torch.ones(i.shape[0], dtype=msg.dtype)).clamp(min=1)
                out = out / cnt.view(-1, *([1] * (msg.dim() - 1)))
            return out

    class GraphNorm(nn.Module):                       # constructed by KarmaDock.__init__, never called by forward()
        def __init__(self, n):
            super().__init__()
            self.weight, self.bias, self.mean_scale = (nn.Parameter(torch.ones(n)), nn.Parameter(torch.zeros(n)), nn.Parameter(torch.ones(n)))

    def to_dense_batch(x, batch, fill_value=0):
        B = int(batch.max()) + 1
        cnt = torch.bincount(batch, minlength=B)
        N = int(cnt.max())
        ptr = torch.cumsum(cnt, 0) - cnt
        pos = torch.arange(batch.shape[0]) - ptr[batch]
        out = x.new_full((B, N) + tuple(x.shape[1:]), fill_value)
        mask = torch.zeros(B, N, dtype=torch.bool)
        out[batch, pos] = x
        mask[batch, pos] = True
        return out, mask

    tg = types.ModuleType("torch_geometric")
    tgn, tgu = types.ModuleType("torch_geometric.nn"), types.ModuleType("torch_geometric.utils")
    tgn.MessagePassing, tgn.GraphNorm = MessagePassing, GraphNorm
    tgu.to_dense_batch = to_dense_batch
    tgu.softmax = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("EGNN is not on the scoring path"))
    tg.nn, tg.utils = tgn, tgu
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tgn, "torch_geometric.utils": tgu})
    ts = types.ModuleType("torch_scatter")
    ts.scatter, ts.scatter_add, ts.scatter_mean = cluster.scatter, cluster.scatter_add, cluster.scatter_mean
    sys.modules["torch_scatter"] = ts


from tests.test_mdn_inputs import mdn_inputs  # noqa: E402


def golden_mdn():
    """f4: the reference's OWN KarmaDock.forward (DiffBindFR/scoring/architecture/*.py executed from source, eval mode)
    on featurised synthetic inputs with seeded weights; asserts the oracle reproduces score and both embeddings; freezes
    inputs + outputs as tests/golden/mdn.npz."""
    print("[f4: MDN scorer network forward]")
    from oracle import mdn_scorer as oms
    import importlib.util
    _pyg_standins()
    base = os.path.join(ref_shims.COPY, "DiffBindFR", "scoring", "architecture")
    pkg = types.ModuleType("dbfr_ref_scoring_arch")
    pkg.__path__ = [base]
    sys.modules["dbfr_ref_scoring_arch"] = pkg
    spec = importlib.util.spec_from_file_location("dbfr_ref_scoring_arch.KarmaDock_sc", os.path.join(base, "KarmaDock_sc.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    torch.manual_seed(0)
    model = mod.KarmaDock().eval()
    P = oms.init_params(seed=3)
    sd = model.state_dict()
    missing = [k for k in P if k not in sd]
    assert not missing, missing
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    used = set(P)
    on_path = [k for k in sd if k.split(".")[0] in ("lig_encoder", "pro_encoder", "mdn_layer") and k not in used
               and not k.endswith("num_batches_tracked") and not k.endswith("dummy_param")
               and not k.startswith(("mdn_layer.atom_types", "mdn_layer.bond_types"))]
    assert not on_path, f"oracle.param_shapes misses tensors of the scoring path: {on_path}"
    model.load_state_dict(P, strict=False)
    rng = np.random.default_rng(12)
    d = mdn_inputs(rng, [(9, 34), (14, 41), (5, 31)])

    class Store(dict):
        __getattr__ = dict.__getitem__

    cov = torch.ones(d["lig_edge_index"].shape[1], dtype=torch.bool)
    data = {"ligand": Store(batch=d["lig_batch"], xyz=d["lig_pos"], node_s=d["lig_node_s"], cov_edge_mask=cov),
            ("ligand", "l2l", "ligand"): Store(edge_s=d["lig_edge_s"], edge_index=d["lig_edge_index"]),
            "protein": Store(node_s=d["pro_node_s"], node_v=d["pro_node_v"], seq=d["pro_seq"], xyz_full=d["pro_xyz_full"], batch=d["pro_batch"]),
            ("protein", "p2p", "protein"): Store(edge_index=d["pro_edge_index"], edge_s=d["pro_edge_s"], edge_v=d["pro_edge_v"])}

    class Hetero(dict):
        def __getitem__(self, k):
            return dict.__getitem__(self, k)
    H = Hetero(data)
    H_get = H.__getitem__
    # HeteroData indexes edge stores by a tuple key: data['ligand', 'l2l', 'ligand'] == data[('ligand','l2l','ligand')]
    with torch.no_grad():
        pro_ref, lig_ref = model.encoding(H)
        score_ref = model(H)
    score, lig_s, pro_s = oms.forward(P, d)
    close(lig_s, lig_ref, 2e-5, "f4/ligand graph-transformer embedding")
    close(pro_s, pro_ref, 2e-5, "f4/pocket GVP embedding")
    close(score, score_ref, 1e-5 * float(score_ref.abs().max()), "f4/MDN score per graph")
    out = {k: npy(v) for k, v in d.items()}
    out.update(params_seed=np.asarray(3), ref_score=npy(score_ref), ref_lig_s=npy(lig_ref), ref_pro_s=npy(pro_ref))
    np.savez_compressed(os.path.join(HERE, "mdn.npz"), **out)
    print("  mdn.npz:", os.path.getsize(os.path.join(HERE, "mdn.npz")) // 1024, "KiB; scores", npy(score_ref))



def golden_mdn_features():
    """f4 (featurisation, pocket half): the reference's own `get_protein_feature`
    (DiffBindFR/scoring/dataset/protein_feature.py:137-216) on the 3DBS pocket of export.npz -- its PDB parser replaced by the
    arrays a parser would return (Bio.PDB absent), everything behind it (openfold transforms the reference vendors, feature
    arithmetic) the reference's code; torch_cluster.knn_graph stood in.  Asserts the oracle and freezes mdn_features.npz."""
    print("[f4: pocket featurisation of the scorer]")
    import importlib, importlib.util
    from oracle import mdn_features as omf
    if "openfold.data.data_transforms" not in sys.modules:
        golden_chi_differ()                                        # installs the openfold stand-ins
    tc = sys.modules["torch_cluster"]
    tc.knn_graph = lambda x, k, **kw: omf.knn_graph(x, k)
    importlib.import_module("openfold.np.protein")
    spec = importlib.util.spec_from_file_location("ref_protein_feature", os.path.join(ref_shims.COPY, "DiffBindFR", "scoring", "dataset", "protein_feature.py"))
    pf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pf)
    z = np.load(os.path.join(HERE, "export.npz"))
    pm = z["pocket_mask"]
    aatype, pos37, m37 = z["aatype"][pm], z["atom37_pos"][pm].astype(np.float32), z["atom37_mask"][pm].astype(np.float32)
    # perturbed side chains so that the centre-of-mass distances are those of a pose, and one residue without its side chain
    rng = np.random.default_rng(3)
    pos37 = pos37 + rng.normal(0, 0.3, pos37.shape).astype(np.float32) * (np.arange(37)[None, :, None] >= 5)
    m37[7, 5:] = 0.0
    pos37 = pos37 * m37[..., None]
    obj = types.SimpleNamespace(aatype=aatype, atom_positions=pos37, atom_mask=m37, residue_index=np.arange(len(aatype)),
                                b_factors=np.zeros_like(m37))
    pf.protein.from_pdb_string = lambda s, c=None: obj
    ca, xyz_full, seq, node_s, node_v, ei, edge_s, edge_v = pf.get_protein_feature("arrays instead of a PDB string", pdb_string=True)
    a14 = torch.from_numpy(T["atom14_to_atom37"][aatype]).long()
    pos14 = torch.from_numpy(pos37)[torch.arange(len(aatype))[:, None], a14] * torch.from_numpy(T["atom14_mask"][aatype]).float()[..., None]
    pos14 = pos14 * torch.from_numpy(m37)[torch.arange(len(aatype))[:, None], a14][..., None]
    close(pos14, xyz_full, 0.0, "f4feat/xyz_full == atom14 of the arrays")
    ideal = torch.from_numpy(T["atom14_mask"][aatype]).sum(-1)
    o = omf.pocket_features(torch.from_numpy(aatype), pos14, ideal, edge_index=ei)
    close(o["node_s"], node_s.float(), 2e-5, "f4feat/node_s (3 distances + 6 backbone dihedral sin/cos)")
    close(o["node_v"], node_v.float(), 2e-6, "f4feat/node_v")
    close(o["edge_s"], edge_s.float(), 2e-5, "f4feat/edge_s (21)")
    close(o["edge_v"], edge_v.float(), 2e-6, "f4feat/edge_v")
    assert torch.equal(omf.knn_graph(ca, 30), ei) and torch.equal(seq, torch.from_numpy(aatype))
    np.savez_compressed(os.path.join(HERE, "mdn_features.npz"), aatype=aatype, atom14_pos=npy(pos14), ideal_atom_count=npy(ideal),
                        ref_node_s=npy(node_s.float()), ref_node_v=npy(node_v.float()), ref_edge_index=npy(ei),
                        ref_edge_s=npy(edge_s.float()), ref_edge_v=npy(edge_v.float()))


# --------------------------------------------------------------------------- boundary: the reference's REAL registry + checkpoint loader
def _ref_function_source(path, start_marker, stop_marker):
    """Source text of one top-level function of a reference file, read at run time (never stored)."""
    src = open(path).read()
    a = src.index(start_marker)
    return src[a: src.index(stop_marker, a)]


def golden_boundary():
    """DiffBindFRHIP / TensorProductModelHIP registered into the reference's OWN registries
    (druglib/utils/registry.py, druglib/models/builder.py), built by its `build_task_model`, and a checkpoint carrying
    e3nn-style buffers + `ema_` duplicates + `module.` prefixes loaded through its OWN `load_checkpoint` /
    `load_state_dict` walk (druglib/core/runner/checkpoint.py:32-100,403-460) with predict.py's arguments
    (strict=True, drop_keys=['^ema_'], DiffBindFR/app/predict.py:118-125).  Freezes the checkpoint's key list / shapes
    as tests/golden/boundary.npz; tests/test_host.py replays it without the reference."""
    print("[boundary: real registry + real checkpoint loader]")
    import importlib.util
    import re
    from collections import OrderedDict
    d = os.path.join(ref_shims.COPY, "druglib")

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(d, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    utils = sys.modules["druglib.utils"]
    load("druglib.utils.misc", "utils/misc.py")
    reg = load("druglib.utils.registry", "utils/registry.py")
    utils.Registry, utils.build_from_cfg = reg.Registry, reg.build_from_cfg        # the REAL ones from here on
    builder = load("druglib.models.builder", "models/builder.py")
    load("druglib.models.Docking.default_MLDockBuilder", "models/Docking/default_MLDockBuilder.py")
    import diffbindfr_amd as dba
    assert dba.register_into_druglib() is True
    assert builder.INTERACTION.get("TensorProductModelHIP") is dba.TensorProductModelHIP
    assert builder.MLDOCK_BUILDER.get("DiffBindFRHIP") is dba.DiffBindFRHIP
    # the keyword of the reference is `overwrite` (registry.py:285-288); the package's mirror takes the same
    try:
        builder.INTERACTION.register_module(name="TensorProductModelHIP", module=dba.TensorProductModelHIP)
        raise AssertionError("re-registration without overwrite must fail")
    except KeyError:
        pass
    builder.INTERACTION.register_module(name="TensorProductModelHIP", overwrite=True, module=dba.TensorProductModelHIP)
    try:
        dba.INTERACTION.register_module(name="TensorProductModelHIP", module=dba.TensorProductModelHIP)
        raise AssertionError("mirror: re-registration without overwrite must fail")
    except KeyError:
        pass
    dba.INTERACTION.register_module(name="TensorProductModelHIP", overwrite=True, module=dba.TensorProductModelHIP)
    # built the way engines.py:147-148 does: cfg.model with --cfg-options model.type=DiffBindFRHIP
    scfg = schedule.default_sample_cfg()
    mcfg_model = ED(type="DiffBindFRHIP", task="mldock", train_cfg=None,
                    diffusion_model=ED(type="TensorProductModelHIP", cfg=ref_model_cfg()))
    model = builder.build_task_model(mcfg_model, test_cfg=ED(sample_cfg=ED(vars(scfg))))
    assert isinstance(model, dba.DiffBindFRHIP) and isinstance(model.diffusion_model, dba.TensorProductModelHIP)
    # a checkpoint as the reference's trainer writes it: the reference model's own keys under `diffusion_model.`,
    # e3nn 0.5.1's persistent buffers where the reference holds e3nn modules (`tp` of every conv, `final_tp_tor`), EMA
    # copies under `ema_` with dots replaced by underscores (hooks/ema.py), everything behind DDP's `module.`
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=5)
    ref_model = ns.tpscore.TensorProductModel(ref_model_cfg())
    assert sorted(ref_model.state_dict()) == sorted(params)
    ck = OrderedDict()
    for k, v in ref_model.state_dict().items():
        ck["module.diffusion_model." + k] = params[k].clone()
    convs = [f"{f}.{l}" for f in ("lig_conv_layers", "atom_conv_layers", "cross_al_conv_layers", "cross_la_conv_layers")
             for l in range(6)] + ["final_conv", "tor_bond_conv", "sc_tor_bond_conv"]
    e3 = []
    for c in convs:
        e3 += [(f"{c}.tp.weight", torch.empty(0)), (f"{c}.tp.output_mask", torch.ones(7)),
               (f"{c}.tp._compiled_main_left_right._w3j_1_1_0", torch.zeros(3, 3, 1))]
    e3 += [("final_tp_tor.weight", torch.empty(0)), ("final_tp_tor.output_mask", torch.ones(45)),
           ("final_tp_tor._compiled_main_left_right._w3j_2_2_1", torch.zeros(5, 5, 3))]
    for k, v in e3:
        ck["module.diffusion_model." + k] = v
    for k in list(ck):
        ck["ema_" + k[len("module."):].replace(".", "_")] = ck[k].clone()
    # the reference's own loader functions, executed from its source
    path = os.path.join(d, "core/runner/checkpoint.py")
    env = {"nn": torch.nn, "OrderedDict": OrderedDict, "Optional": __import__("typing").Optional, "logging": __import__("logging"),
           "Union": __import__("typing").Union, "Any": __import__("typing").Any, "List": list, "Tuple": tuple, "re": re,
           "get_dist_info": lambda: (0, 1)}
    preg = types.ModuleType("parallel_registry")
    preg.MODULE_WRAPPERS = reg.Registry("module wrapper")
    preg.MODULE_WRAPPERS.register_module(module=torch.nn.parallel.DataParallel)
    preg.MODULE_WRAPPERS.register_module(module=torch.nn.parallel.DistributedDataParallel)
    penv = {"nn": torch.nn, "MODULE_WRAPPERS": preg.MODULE_WRAPPERS}
    exec(_ref_function_source(os.path.join(d, "core/runner/parallel/utils.py"), "def is_module_wrapper(", "\ndef "), penv)
    env["is_module_wrapper"] = penv["is_module_wrapper"]
    exec(_ref_function_source(path, "def load_state_dict(", "\ndef get_torchvision_models"), env)
    exec(_ref_function_source(path, "\ndef load_checkpoint(\n        model", "\ndef weights_to_cpu"), env)
    env["_load_checkpoint"] = lambda filename, map_location, logger: {"state_dict": ck, "meta": {}}
    env["load_checkpoint"](model, "diffbindfr_paper.pth", map_location="cpu", strict=True, logger=None,
                           revise_keys=[(r"^module\.", "")], drop_keys=[r"^ema_"], use_ema=False)
    got = model.diffusion_model.state_dict()
    for k, v in params.items():
        assert torch.equal(got[k], v), k
    absorbed = sorted(model.diffusion_model.ignored_keys)
    assert absorbed == sorted("diffusion_model." + k for k, _ in e3), "the e3nn stand-ins must take exactly the e3nn keys"
    print(f"  reference load_checkpoint(strict=True): {len(params)} tensors loaded, {len(absorbed)} e3nn buffers absorbed")
    # a genuinely unexpected / missing key must still fail under strict
    for bad in ("extra", "missing"):
        ck2 = OrderedDict(ck)
        if bad == "extra":
            ck2["module.diffusion_model.lig_conv_layers.0.fc.lin.9.weight"] = torch.zeros(1)
        else:
            del ck2["module.diffusion_model.final_conv.fc.lin.0.bias"]
        env["_load_checkpoint"] = lambda filename, map_location, logger, c=ck2: {"state_dict": c}
        try:
            env["load_checkpoint"](model, "x.pth", strict=True, drop_keys=[r"^ema_"])
            raise AssertionError(f"strict load must reject a checkpoint with an {bad} key")
        except RuntimeError:
            pass
    keys = [k for k in ck]
    np.savez_compressed(os.path.join(HERE, "boundary.npz"),
                        keys=np.asarray(keys), shapes=np.asarray([",".join(map(str, ck[k].shape)) for k in keys]),
                        absorbed=np.asarray(absorbed), n_params=np.asarray(len(params)))



if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    if sys.argv[1:] == ["sampler_modes"]:
        golden_sampler_modes()
        sys.exit(0)
    if sys.argv[1:] == ["examples"]:          # the two example fixtures alone (15 + 6 poses through the reference's sample(): minutes)
        golden_examples()
        sys.exit(0)
    golden_geometry()
    golden_embeddings()
    golden_schedule()
    golden_model_and_sampler()
    golden_sampler_modes()
    golden_pose_init()
    golden_pocket()
    golden_real_complex()
    golden_real_trajectory()
    golden_examples()
    golden_export()
    golden_pocket_select()
    golden_chi_differ()
    golden_mdn()
    golden_mdn_features()
    golden_boundary()           # last: swaps the stand-in registry for the reference's real one
    print("golden fixtures written to", HERE)
