"""Row f4: the MDN pose scorer's network forward -- KarmaDock.forward of
DiffBindFR/scoring/architecture/KarmaDock_sc.py (graph-transformer ligand encoder, GVP pocket encoder, mixture-density
head).  tests/golden/mdn.npz holds featurised synthetic inputs and what the REFERENCE's own architecture files computed
for them with seeded weights (tests/golden/make_golden.py::golden_mdn).  CPU: the oracle against the fixture.  GPU: the
HIP path (dbfr_mdn_*) against the fixture and against the oracle on ragged batches."""
import os

import numpy as np
import pytest
import torch

from oracle import mdn_scorer as oms
from tests.helpers import GOLDEN, rel_err


def fixture():
    z = np.load(os.path.join(GOLDEN, "mdn.npz"))
    d = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith(("ref_", "params_"))}
    return d, z


def test_oracle_matches_reference_fixture():
    d, z = fixture()
    P = oms.init_params(seed=int(z["params_seed"]))
    score, lig_s, pro_s = oms.forward(P, d)
    assert (lig_s - torch.from_numpy(z["ref_lig_s"])).abs().max() < 2e-5
    assert (pro_s - torch.from_numpy(z["ref_pro_s"])).abs().max() < 2e-5
    assert rel_err(score, torch.from_numpy(z["ref_score"])) < 1e-5


def test_pair_distance_quirks():
    """MDN_Block.compute_euclidean_distances_matrix: float64 expansion |x|^2 + |y|^2 - 2xy, NaN -> 10000, min over the 14
    slots with unused slots (zeros) NOT masked."""
    lig = torch.tensor([[1.0, 2.0, 3.0], [50.0, 0.0, 0.0]])
    res = torch.zeros(2, 14, 3)
    res[0, 0] = torch.tensor([1.0, 2.0, 3.0])          # coincident with ligand atom 0: distance 0 (or 10000 if d^2 < 0)
    res[0, 1:] = 100.0
    res[1, :2] = torch.tensor([[50.0, 3.0, 4.0], [60.0, 0.0, 0.0]])   # slots 2.. stay at the origin
    d = oms.pair_distance(lig, res)
    assert d.dtype == torch.float64 and d.shape == (2, 2)
    assert d[0, 0] in (0.0, 10000.0) and abs(d[1, 1] - 5.0) < 1e-9
    assert abs(d[0, 1] - (14.0 ** 0.5)) < 1e-9           # the origin slots of residue 1 are nearer than its real atoms


def test_product_parameter_names_match_the_oracle():
    from diffbindfr_amd import mdn
    assert mdn.param_shapes() == {k: tuple(v) for k, v in oms.param_shapes().items()}


# ------------------------------------------------------------------------------------------------ GPU
def _hip_model(dev, seed):
    from diffbindfr_amd import mdn
    P = oms.init_params(seed=seed)
    m = mdn.KarmaDockHIP().to(dev)
    m.load_state_dict(P, strict=True)
    return P, m


def _to(d, dev):
    return {k: v.to(dev) for k, v in d.items()}


@pytest.mark.gpu
def test_gpu_scorer_matches_reference_fixture():
    """Scores and both embeddings of the HIP forward against what the reference's KarmaDock computed (mdn.npz).
    Tolerance: embeddings 1e-4 absolute (fp32 sums in another order), scores 1e-4 relative."""
    dev = torch.device("cuda:0")
    d, z = fixture()
    P, m = _hip_model(dev, int(z["params_seed"]))
    score, lig_s, pro_s = m.score(_to(d, dev), return_embeddings=True)
    torch.cuda.synchronize()
    assert (lig_s.cpu() - torch.from_numpy(z["ref_lig_s"])).abs().max() < 1e-4
    assert (pro_s.cpu() - torch.from_numpy(z["ref_pro_s"])).abs().max() < 1e-4
    assert rel_err(score, torch.from_numpy(z["ref_score"])) < 1e-4, (score.cpu(), z["ref_score"])


@pytest.mark.gpu
def test_gpu_scorer_ragged_batches_vs_oracle_and_reuse_of_ligand_embeddings():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    dev = torch.device("cuda:0")
    P, m = _hip_model(dev, 7)
    rng = np.random.default_rng(5)
    from tests.test_mdn_inputs import mdn_inputs
    for sizes in ([(4, 31)], [(30, 140), (6, 33), (21, 75), (12, 90)], [(35, 105)] * 3):
        d = mdn_inputs(rng, sizes, coincident=False)
        ref, lig_ref, pro_ref = oms.forward(P, d)
        score, lig_s, pro_s = m.score(_to(d, dev), return_embeddings=True)
        assert (lig_s.cpu() - lig_ref).abs().max() < 1e-4 and (pro_s.cpu() - pro_ref).abs().max() < 1e-4
        assert rel_err(score, ref) < 1e-4, (sizes, score.cpu(), ref)
        # pose-independent ligand embeddings handed back in: same scores, bit for bit
        again = m.score(_to(d, dev), lig_s=lig_s)
        assert torch.equal(again, score)
        assert torch.equal(m.score(_to(d, dev)), score)            # reproducible
    # graph order does not matter
    d = mdn_inputs(rng, [(9, 40), (15, 52)], coincident=False)
    a = m.score(_to(d, dev)).cpu()
    sw = mdn_inputs(np.random.default_rng(99), [(15, 52)], coincident=False)   # unrelated batch in between
    m.score(_to(sw, dev))
    assert torch.equal(m.score(_to(d, dev)).cpu(), a)


@pytest.mark.gpu
def test_gpu_scorer_refuses_cpu_tensors_and_bad_state():
    from diffbindfr_amd import lib as L, mdn
    d, z = fixture()
    m = mdn.KarmaDockHIP()
    with pytest.raises(L.DbfrError):
        m.score(d)


def test_oracle_pocket_features_match_reference_fixture():
    """mdn_features.npz: the reference's own get_protein_feature on the 3DBS pocket (perturbed side chains, one residue
    without its side chain)."""
    from oracle import mdn_features as omf
    z = np.load(os.path.join(GOLDEN, "mdn_features.npz"))
    o = omf.pocket_features(torch.from_numpy(z["aatype"]), torch.from_numpy(z["atom14_pos"]), torch.from_numpy(z["ideal_atom_count"]))
    assert torch.equal(o["edge_index"], torch.from_numpy(z["ref_edge_index"]))
    assert (o["node_s"] - torch.from_numpy(z["ref_node_s"])).abs().max() < 2e-5
    assert (o["node_v"] - torch.from_numpy(z["ref_node_v"])).abs().max() < 1e-6
    assert (o["edge_s"] - torch.from_numpy(z["ref_edge_s"])).abs().max() < 1e-6
    assert (o["edge_v"] - torch.from_numpy(z["ref_edge_v"])).abs().max() < 1e-6


@pytest.mark.gpu
def test_gpu_pocket_features_match_reference_fixture_and_oracle():
    from diffbindfr_amd import mdn
    from oracle import mdn_features as omf
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, "mdn_features.npz"))
    aa, x = torch.from_numpy(z["aatype"]), torch.from_numpy(z["atom14_pos"])
    f = mdn.pocket_features(aa.to(dev), x.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(f["pro_edge_index"].cpu(), torch.from_numpy(z["ref_edge_index"]))
    assert (f["pro_node_s"].cpu() - torch.from_numpy(z["ref_node_s"])).abs().max() < 5e-5
    assert (f["pro_node_v"].cpu() - torch.from_numpy(z["ref_node_v"])).abs().max() < 1e-5
    assert (f["pro_edge_s"].cpu() - torch.from_numpy(z["ref_edge_s"])).abs().max() < 1e-5
    assert (f["pro_edge_v"].cpu() - torch.from_numpy(z["ref_edge_v"])).abs().max() < 1e-5
    # several poses of several pockets in one launch (ragged), incl. a pocket smaller than k + 1
    rng = np.random.default_rng(1)
    sizes = [105, 12, 60]
    parts, ptr = [], [0]
    for n in sizes:
        sel = rng.choice(105, n, replace=False)
        sel.sort()
        parts.append((aa[sel], x[sel] + torch.from_numpy(rng.normal(0, 0.2, (n, 14, 3))).float() * (x[sel].abs().sum(-1, keepdim=True) > 0)))
        ptr.append(ptr[-1] + n)
    A, X = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    f = mdn.pocket_features(A.to(dev), X.to(dev), res_ptr=ptr)
    e0 = 0
    for (a_, x_), n in zip(parts, sizes):
        ideal = torch.tensor([5, 11, 8, 8, 6, 9, 9, 4, 10, 8, 8, 9, 8, 11, 7, 6, 7, 14, 12, 7])[a_]
        o = omf.pocket_features(a_, x_, ideal)
        ne = o["edge_index"].shape[1]
        r0 = ptr[sizes.index(n)]                                   # sizes are distinct
        assert torch.equal(f["pro_edge_index"][:, e0:e0 + ne].cpu() - r0, o["edge_index"])
        assert (f["pro_edge_s"][e0:e0 + ne].cpu() - o["edge_s"]).abs().max() < 1e-5
        assert (f["pro_node_s"][r0:r0 + n].cpu() - o["node_s"]).abs().max() < 5e-5
        e0 += ne
    # the features feed the scorer directly
    P, m = _hip_model(dev, 7)
    d = mdn_small_batch(rng, f, ptr)
    s = m.score(d)
    assert torch.isfinite(s).all() and s.shape == (3,)


def mdn_small_batch(rng, f, ptr):
    """Ligand halves (synthetic features) for the three pockets of the test above."""
    from tests.test_mdn_inputs import mdn_inputs
    dev = f["pro_node_s"].device
    lig = mdn_inputs(rng, [(8, 20), (5, 20), (11, 20)], coincident=False)
    d = {k: v.to(dev) for k, v in lig.items() if k.startswith("lig_")}
    d.update(f)
    return d


@pytest.mark.gpu
def test_gpu_score_poses_of_one_complex_vs_oracle():
    """P poses of one complex straight from (aatype, atom14 poses, ligand poses): device pocket features + one ligand-encoder
    pass + the mixture head, against the oracle's featurisation + forward pose by pose."""
    from diffbindfr_amd import mdn
    from oracle import mdn_features as omf
    from tests.test_mdn_inputs import mdn_inputs
    dev = torch.device("cuda:0")
    P_, m = _hip_model(dev, 11)
    z = np.load(os.path.join(GOLDEN, "mdn_features.npz"))
    rng = np.random.default_rng(2)
    aa, x0 = torch.from_numpy(z["aatype"]), torch.from_numpy(z["atom14_pos"])
    lig = {k: v for k, v in mdn_inputs(rng, [(17, 20)], coincident=False).items() if k in ("lig_node_s", "lig_edge_s", "lig_edge_index")}
    n_pose = 5
    present = (x0.abs().sum(-1, keepdim=True) > 0).float()
    poses = torch.stack([x0 + torch.from_numpy(rng.normal(0, 0.4, x0.shape)).float() * present * (torch.arange(14)[None, :, None] >= 5) for _ in range(n_pose)])
    ca = x0[:, 1]
    lpos = torch.stack([ca[rng.integers(0, len(ca), 17)] + torch.from_numpy(rng.normal(0, 2.0, (17, 3))).float() for _ in range(n_pose)])
    got = m.score_poses(lig, aa, poses.to(dev), lpos.to(dev)).cpu()
    again = m.score_poses(lig, aa, poses.to(dev), lpos.to(dev), reuse_ligand_embeddings=False).cpu()
    assert torch.equal(got, again)
    # several complexes in one call: same scores as one by one
    sub = slice(10, 70)
    many = m.score_complexes([(lig, aa, poses.to(dev), lpos.to(dev)), (lig, aa[sub], poses[:3, sub].to(dev), lpos[:3].to(dev))])
    assert torch.equal(many[0].cpu(), got)
    assert torch.equal(many[1].cpu(), m.score_poses(lig, aa[sub], poses[:3, sub].to(dev), lpos[:3].to(dev), reuse_ligand_embeddings=False).cpu())
    ideal = torch.from_numpy(z["ideal_atom_count"])
    for p in range(n_pose):
        f = omf.pocket_features(aa, poses[p], ideal)
        d = dict(lig, lig_pos=lpos[p], lig_batch=torch.zeros(17, dtype=torch.long), pro_node_s=f["node_s"], pro_node_v=f["node_v"],
                 pro_edge_index=f["edge_index"], pro_edge_s=f["edge_s"], pro_edge_v=f["edge_v"], pro_seq=aa, pro_xyz_full=f["xyz_full"],
                 pro_batch=torch.zeros(len(aa), dtype=torch.long))
        ref, _, _ = oms.forward(P_, d)
        assert abs(float(got[p]) - float(ref[0])) <= 2e-4 * max(1.0, abs(float(ref[0]))), (p, got[p], ref[0])


# ------------------------------------------------------------------------------------------------ Scorer entry / registry (row f4 glue)
def test_collate_flat_equals_a_batch_built_at_once():
    """``mdn.collate_flat`` (the PyG DataLoader's job in engines.py:270-277): per-pair flat dicts -> one batch with shifted edge
    indices and renumbered graph ids; equal to generating the batch at once, and the oracle scores it the same."""
    from diffbindfr_amd import mdn
    from tests.test_mdn_inputs import mdn_inputs
    sizes = [(9, 40), (15, 52), (4, 31)]
    whole = mdn_inputs(np.random.default_rng(3), sizes, coincident=False)
    lb, pb = whole["lig_batch"], whole["pro_batch"]
    items = []
    for g in range(len(sizes)):
        lm, pm = lb == g, pb == g
        l0, p0 = int(torch.nonzero(lm)[0]), int(torch.nonzero(pm)[0])
        le = lm[whole["lig_edge_index"][0]]
        pe = pm[whole["pro_edge_index"][0]]
        it = {k: whole[k][lm] for k in ("lig_node_s", "lig_pos")}
        it.update({k: whole[k][pm] for k in ("pro_node_s", "pro_node_v", "pro_seq", "pro_xyz_full")})
        it.update(lig_edge_s=whole["lig_edge_s"][le], lig_edge_index=whole["lig_edge_index"][:, le] - l0,
                  pro_edge_s=whole["pro_edge_s"][pe], pro_edge_v=whole["pro_edge_v"][pe], pro_edge_index=whole["pro_edge_index"][:, pe] - p0)
        items.append(it)
    got = mdn.collate_flat(items)
    assert set(got) == set(whole)
    for k in whole:
        assert torch.equal(got[k], whole[k]), k


def test_scoring_model_goes_through_the_energy_registry():
    """scFlex.py:43-46: ``DiffBindFR(scoring_model=cfg)`` builds it with ``build_energy``; here ENERGY holds KarmaDockHIP
    ('KarmaDock' as a type name is mapped onto it)."""
    import diffbindfr_amd as dba
    from diffbindfr_amd import mdn, registry
    assert registry.ENERGY.get("KarmaDockHIP") is mdn.KarmaDockHIP
    for cfg in ({"type": "KarmaDockHIP"}, {"type": "KarmaDock", "cfg": {"x": 1}}):
        m = dba.DiffBindFRHIP(diffusion_model=None, scoring_model=cfg)
        assert isinstance(m.scoring_model, mdn.KarmaDockHIP) and not hasattr(m, "diffusion_model")
    both = dba.DiffBindFRHIP(diffusion_model={"type": "TensorProductModel", "cfg": {}}, scoring_model={"type": "KarmaDockHIP"})
    assert isinstance(both.scoring_model, mdn.KarmaDockHIP) and isinstance(both.diffusion_model, dba.TensorProductModelHIP)
    with pytest.raises(KeyError):
        dba.DiffBindFRHIP(diffusion_model=None, scoring_model={"type": "NoSuchScorer"})


def test_scorer_checkpoint_layouts_of_the_reference_loader():
    """DiffBindFR/scoring/utils/early_stop.py:27-38: `Scorer` loads `mdn_paper.pt` with `mine=True` = `torch.load(f)['model']`, every key
    re-keyed as 'module.' + k[6:]; `mine=False` reads ['model_state_dict'].  All three layouts (+ DataParallel's prefix) must reach the
    model's own names -- and a checkpoint that reaches none of them must raise instead of scoring with the initial parameters."""
    from diffbindfr_amd import lib as L, mdn
    P = oms.init_params(seed=11)
    model = mdn.KarmaDockHIP()
    assert set(model.state_dict()) <= set(P) | set(model.state_dict())
    before = {k: v.clone() for k, v in model.state_dict().items()}
    extra = {"egnn_layers.0.weight": torch.zeros(3)}            # the checkpoint carries modules the scoring forward never calls
    for ckpt in ({"model": {"model." + k: v for k, v in {**P, **extra}.items()}},          # six-character wrapper prefix
                 {"model": {"abcde." + k: v for k, v in P.items()}},                      # (any six characters: the reference cuts by count)
                 {"model_state_dict": {"module." + k: v for k, v in P.items()}},
                 {"module." + k: v for k, v in P.items()}, dict(P)):
        m = mdn.KarmaDockHIP()
        n = mdn.load_scorer_weights(m, ckpt)
        assert n == len(m.state_dict())
        for k, v in m.state_dict().items():
            assert torch.equal(v, P[k].to(v.dtype)), k
    assert any(not torch.equal(before[k], P[k].to(before[k].dtype)) for k in before)      # (the load changed something)
    for bad in ({"model": dict(P)},                              # 'model' layout without the prefix: six characters of every NAME are cut
                {"state_dict": dict(P)}, {"model." + k: v for k, v in P.items()}):
        with pytest.raises(L.DbfrError, match="refusing to score"):
            mdn.load_scorer_weights(mdn.KarmaDockHIP(), bad)
    with pytest.raises(KeyError, match="collate_flat"):
        mdn.collate_flat([{"lig_pos": torch.zeros(2, 3), "pro_node_s": torch.zeros(3, 9)},
                          {"lig_pos": torch.zeros(2, 3), "pro_node_s": torch.zeros(3, 9), "lig_batch": torch.zeros(2)}])


@pytest.mark.gpu
def test_gpu_scorer_entry_writes_the_pair_frame(tmp_path):
    """``mdn.Scorer`` (engines.py:230-302): samples one by one -> batches of `batch_size` -> mdn_score column + csv; the scores
    equal the oracle's on the same samples, whatever the batch size, and DataParallel's `module.` key prefix is accepted."""
    import pandas as pd
    from diffbindfr_amd import mdn
    from tests.test_mdn_inputs import mdn_inputs
    dev = torch.device("cuda:0")
    P = oms.init_params(seed=11)
    rng = np.random.default_rng(8)
    items = [mdn_inputs(rng, [s], coincident=False) for s in ((9, 40), (15, 52), (4, 31), (21, 75), (12, 60))]
    ref = torch.cat([oms.forward(P, it)[0] for it in items])

    class DS(list):
        pass
    out = []
    for bs in (2, 16):
        ds = DS(items)
        ds.pair_frame = pd.DataFrame({"pair": list(range(len(items)))})
        csv = tmp_path / f"mdn_{bs}.csv"
        s = mdn.Scorer(ds, model_weight={"module." + k: v for k, v in P.items()}, output_path=csv, batch_size=bs, device_id=0)
        assert rel_err(torch.tensor(s), ref) < 1e-4
        back = pd.read_csv(csv)
        assert list(back.columns) == ["pair", "mdn_score"] and np.allclose(back["mdn_score"], s)
        out.append(s)
    assert np.allclose(out[0], out[1], rtol=1e-5)
    # the checkpoint layout `Scorer` meets in the reference (mdn_paper.pt through Early_stopper.load_model(mine=True)), and a sample whose
    # featurisation failed (None): left out of its batch like PassNoneDataLoader does, NaN in its row
    ds = DS(items[:2] + [None] + items[2:])
    ds.pair_frame = pd.DataFrame({"pair": list(range(len(ds)))})
    s = mdn.Scorer(ds, model_weight={"model": {"model." + k: v for k, v in P.items()}}, output_path=tmp_path / "mdn_none.csv", batch_size=4, device_id=0)
    assert len(s) == 6 and np.isnan(s[2]) and rel_err(torch.tensor(s[:2] + s[3:]), ref) < 1e-4
    from diffbindfr_amd import lib as L
    with pytest.raises(L.DbfrError, match="refusing to score"):
        mdn.Scorer(DS(items), model_weight={"model": dict(P)}, output_path=None, device_id=0)
