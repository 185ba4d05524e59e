"""BASELINE config 1's structure (examples/forward/3dbs_protein.pdb + crystal ligand, 12 A pocket: 105 residues,
866 heavy atoms, 35 ligand atoms -- the counts SURVEY.md section 8 quotes) as a fixture frozen from the reference's own
functions (tests/golden/real_3dbs.npz, see make_golden.py: golden_real_complex).

CPU: oracle and product host code against the reference outputs, edge counts at the crystal pose (9 002 pocket
edges, 458 ligand radius edges as in SURVEY.md).  GPU: template extraction on real side chains, and the whole chain
raw coordinates -> records -> assembled batch -> score network against the oracle at the real size.
"""
import copy
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import cluster, ligand as olig, pocket as opk, pose_init as opi
from diffbindfr_amd import assemble, ligand as plig, pocket, synthetic
from tests.helpers import GOLDEN, rel_err

T = synthetic.residue_tables()


def fixture():
    return np.load(os.path.join(GOLDEN, "real_3dbs.npz"))


def test_counts_match_survey():
    z = fixture()
    assert z["aatype"].shape[0] == 105 and int(z["atom14_mask"].sum()) == 866 and z["lig_pos"].shape[0] == 35
    assert int(z["atom14_mask"][:, 1].sum() + z["atom14_mask"][:, 4].sum()) == 205
    rec = torch.from_numpy(z["atom14_position"][z["atom14_mask"]]).float()
    e_aa = cluster.radius_graph(rec, 4.0, torch.zeros(866, dtype=torch.long), max_num_neighbors=1000).shape[1]
    e_ll = cluster.radius_graph(torch.from_numpy(z["lig_pos"]).float(), 5.0, torch.zeros(35, dtype=torch.long), max_num_neighbors=32).shape[1]
    assert (e_aa, e_ll) == (9002, 458) == (int(z["e_aa"]), int(z["e_ll"]))


def test_oracle_pocket_functions_on_real_side_chains():
    z = fixture()
    seq = z["aatype"]
    ideal = T["atom14_mask"][seq][..., None].astype(np.float32)
    mine = opk.extract_chi_and_template(seq, z["atom14_position"].copy(), ideal, T)
    for k in ("backbone_transl", "backbone_rots", "default_frame", "rigid_group_positions", "torsion_angle"):
        assert np.abs(np.asarray(mine[k], np.float64) - z["ref_" + k]).max() <= 1e-6, k
    te, cm = opk.build_torsion_edges(torch.from_numpy(seq), torch.from_numpy(z["atom14_mask"]), T)
    assert torch.equal(te, torch.from_numpy(z["ref_torsion_edge_index"])) and torch.equal(cm, torch.from_numpy(z["ref_sc_torsion_edge_mask"]))


def test_torsion_finder_on_the_real_ligand_and_random_molecules():
    z = fixture()
    for fn in (olig.find_torsion, plig.torsion_masks):
        tor, rot = fn(35, z["lig_edge_index"])
        assert np.array_equal(tor, z["ref_tor_edge_mask"]) and np.array_equal(rot, z["ref_rot_node_mask"]), fn.__module__
    assert int(z["ref_tor_edge_mask"].sum()) >= 3 and z["lig_edge_index"].shape[1] == 80      # rings: 40 bonds on 35 atoms
    rng = np.random.default_rng(2)
    for _ in range(25):                                         # product == oracle, incl. equal-halves ties (n even, chains)
        lg = synthetic.make_ligand(rng, int(rng.integers(2, 30)))
        a, b = olig.find_torsion(lg["n_lig"], lg["lig_edge_index"]), plig.torsion_masks(lg["n_lig"], lg["lig_edge_index"])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    chain = np.array([[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2]])  # 4-chain: the middle bond splits 2 | 2
    a, b = olig.find_torsion(4, chain), plig.torsion_masks(4, chain)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and int(a[0].sum()) == 1


def _ligand_half(z, seed=0):
    """Real geometry and bond graph, seeded synthetic features (the RDKit featuriser is outside this path)."""
    rng = np.random.default_rng(seed)
    ei = z["lig_edge_index"]
    node = np.clip(rng.standard_normal((35, 27)), -3, 3).astype(np.float32)
    und = {}
    feat = np.zeros((ei.shape[1], 10), np.float32)
    for k, (u, v) in enumerate(ei.T.tolist()):
        key = (min(u, v), max(u, v))
        if key not in und:
            und[key] = (int(rng.integers(0, 4)), rng.integers(0, 2, 4))
        feat[k, und[key][0]] = 1.0
        feat[k, 6:] = und[key][1]
    return dict(lig_node=torch.from_numpy(node), lig_edge_index=torch.from_numpy(ei), lig_edge_feat=torch.from_numpy(feat),
                tor_edge_mask=torch.from_numpy(z["ref_tor_edge_mask"]).long(), rot_node_mask=torch.from_numpy(z["ref_rot_node_mask"]))


@pytest.mark.gpu
def test_gpu_templates_of_real_side_chains():
    dev = torch.device("cuda:0")
    z = fixture()
    out = pocket.extract_templates(torch.from_numpy(z["aatype"]).to(dev), torch.from_numpy(z["atom14_position"]).float().to(dev))
    torch.cuda.synchronize()
    for k in ("backbone_rots", "default_frame", "rigid_group_positions"):
        err = np.abs(out[k].cpu().numpy().astype(np.float64) - z["ref_" + k]).max()
        assert err <= 1e-4, (k, err)
    assert np.abs(out["backbone_transl"].cpu().numpy() - z["ref_backbone_transl"]).max() <= 1e-5     # CA positions, ~60 A
    d = np.abs(np.angle(np.exp(1j * (out["torsion_angle"].cpu().numpy().astype(np.float64) - z["ref_torsion_angle"]))))
    assert d.max() <= 3e-4


@pytest.mark.gpu
def test_gpu_real_complex_scores_match_oracle():
    """3DBS pocket + crystal ligand at the crystal pose: coordinates -> pocket_records -> ComplexRecord -> assemble ->
    dbfr_score, against the oracle forward on the reference-format batch; device edge counts == the oracle's."""
    import ctypes as C
    from diffbindfr_amd import lib as L
    from oracle import sampler as osampler, schedule as osched, score_model as sm
    dev = torch.device("cuda:0")
    z = fixture()
    half = pocket.pocket_records(z["aatype"], z["atom14_position"], z["atom14_mask"], None, dev)[0]
    lig = _ligand_half(z)
    lig["lig_pos"] = torch.from_numpy(z["lig_pos"]).float() - half["pocket_center_pos"]          # Decentration moves the ligand too
    rec = {**half, **lig}
    pb = assemble.assemble([assemble.ComplexRecord(copy.deepcopy(rec))], 1, dev)
    # crystal pose, input side chains: rec_pos = atom14_position[mask] (no randomisation)
    m = half["atom14_mask"]
    pb.t["rec_pos"].copy_(half["atom14_position"][m].to(dev))
    import diffbindfr_amd as dba
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=1)
    model = dba.TensorProductModelHIP({}).to(dev)
    model.load_state_dict(params, strict=True)
    # reference-format batch for the oracle
    pose = dict(rec)
    pose["rec_atm_pos"] = half["atom14_position"][m]
    pose["pocket_node_feature"] = half["pocket_node_feature"][m]
    d = SimpleNamespace(**opi.collate([pose]))
    d.batch = d.lig_node_batch
    sc = osched.step_scalars(osched.default_sample_cfg(), 12)
    dd = osampler.set_time(copy.deepcopy(d), sc, 1)
    ref = sm.forward(params, mcfg, copy.deepcopy(dd))
    scn = dd.sc_tor_score_norm2
    scm = dd.sc_torsion_edge_mask.bool()
    out = model.score_packed(pb, dd.t, dd.tr_sigma, dd.rot_score_norm, dd.tor_score_norm2, scn[scm] if scn.shape == scm.shape else scn)
    for nm, a, b in zip(("tr", "rot", "tor", "sc_tor"), out, ref):
        assert rel_err(a, b) < 1e-4, (nm, rel_err(a, b))
    counters = (C.c_int64 * 8)()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(L.load().dbfr_status_sync(C.c_void_p(model.workspace_of(dev).data_ptr()), stream, counters))
    assert counters[1] == 9002 and counters[0] == 458 + 80          # pocket edges; ligand radius + bond edges


@pytest.mark.gpu
def test_gpu_real_complex_20_step_trajectory_matches_the_reference():
    """BASELINE config 1's structure end to end: tests/golden/real_3dbs_traj.npz holds the batch (3DBS pocket built by
    the reference's own pocket pipeline + crystal ligand, 2 initial poses), the noise tape and the trajectory the
    REFERENCE's `DiffBindFR.sample()` produced for it (make_golden.py: golden_real_trajectory).  The HIP sampler must
    follow it for all 20 steps within 1e-3 A."""
    import diffbindfr_amd as dba
    from diffbindfr_amd.packing import PackedBatch
    from oracle import score_model as sm
    from tests.helpers import load_golden_batch, namespace_to
    dev = torch.device("cuda:0")
    d, z = load_golden_batch(os.path.join(GOLDEN, "real_3dbs_traj.npz"))
    assert int(d.rec_atm_pos.shape[0]) == 2 * 866 and int(d.lig_pos.shape[0]) == 2 * 35
    params = sm.init_params(sm.default_cfg(), seed=int(z["params_seed"]))
    model = dba.TensorProductModelHIP({}).to(dev)
    model.load_state_dict(params, strict=True)
    samp = dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})
    pb = PackedBatch(namespace_to(d, dev), dev)
    noise = {k: torch.from_numpy(z[f"noise_{k}"]).to(dev).contiguous() for k in ("tr", "rot", "tor", "sc")}
    lig, a14 = samp.sample_packed(pb, noise, visualize=True)
    dl = (lig.cpu() - torch.from_numpy(z["traj_lig"])).norm(dim=-1)
    assert dl.max() < 1e-3, float(dl.max())
    assert (a14[0].cpu() - torch.from_numpy(z["atom14_step0"])).norm(dim=-1).max() < 1e-3
    assert (a14[-1].cpu() - torch.from_numpy(z["final_atom14"])).norm(dim=-1).max() < 1e-3


def test_oracle_reproduces_the_reference_trajectory_head():
    """CPU: the first two steps of the same fixture through the oracle (the full 20 steps take minutes on few cores;
    make_golden.py asserted all 20 at max|d| = 0)."""
    from oracle import sampler as osampler, schedule as osched, score_model as sm
    from tests.helpers import load_golden_batch
    d, z = load_golden_batch(os.path.join(GOLDEN, "real_3dbs_traj.npz"))
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=int(z["params_seed"]))
    noise = SimpleNamespace(**{k: torch.from_numpy(z[f"noise_{k}"]) for k in ("tr", "rot", "tor", "sc")})
    lig, a14 = osampler.sample(params, mcfg, osched.default_sample_cfg(actual_steps=2), copy.deepcopy(d), noise,
                               torch.from_numpy(T["atom14_to_group"]).long(), visualize=True)
    assert (lig - torch.from_numpy(z["traj_lig"][:2])).abs().max() < 1e-5
    assert (a14[0] - torch.from_numpy(z["atom14_step0"])).abs().max() < 1e-5
