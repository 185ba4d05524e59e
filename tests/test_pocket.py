"""once-per-pocket preparation.

CPU: the oracle (oracle/pocket.py) against the fixture frozen from the reference's own extract_chi_and_template /
make_torsion_mask / build_torsion_edges / PocketFeaturizer (tests/golden/pocket.npz); the product's table look-ups
(diffbindfr_amd/pocket.py) against the oracle.  GPU: dbfr_extract_templates against the reference fixture, the
round trip through the side-chain rebuild, and records built from raw coordinates through the whole sampler.
"""
import os

import numpy as np
import pytest
import torch

from oracle import geometry, pocket as opk
from diffbindfr_amd import assemble, pocket, synthetic
from tests.helpers import GOLDEN

T = synthetic.residue_tables()
TOL = 1e-4      # Angstrom / radian; local coordinates are < 15 A, fp32 on the device vs float64 in the reference


def fixture():
    return np.load(os.path.join(GOLDEN, "pocket.npz"))


def ang_diff(a, b):
    return np.abs(np.angle(np.exp(1j * (np.asarray(a, np.float64) - np.asarray(b, np.float64)))))


def test_oracle_extract_matches_reference_fixture():
    z = fixture()
    mine = opk.extract_chi_and_template(z["aatype"], z["atom14_position"].copy(), z["ideal_mask"][..., None], T)
    for k in ("backbone_transl", "backbone_rots", "default_frame", "rigid_group_positions", "torsion_angle"):
        assert np.abs(np.asarray(mine[k], np.float64) - z["ref_" + k]).max() <= 1e-6, k
    assert set(z["aatype"].tolist()) == set(range(20))       # every residue type is exercised


def test_oracle_masks_edges_features_match_reference_fixture():
    z = fixture()
    seq, actual = torch.from_numpy(z["aatype"]), torch.from_numpy(z["actual_mask"])
    te, cm = opk.build_torsion_edges(seq, actual, T)
    assert torch.equal(te, torch.from_numpy(z["ref_torsion_edge_index"]))
    assert torch.equal(cm, torch.from_numpy(z["ref_sc_torsion_edge_mask"]))
    assert torch.equal(opk.pocket_features(seq, actual, T), torch.from_numpy(z["ref_pocket_node_feature"]))
    assert int((~cm & torch.from_numpy(T["chi_mask"][z["aatype"]]).bool()).sum()) > 0    # missing side chains really drop chis


def test_product_tables_match_oracle_single_and_multi_pocket():
    z = fixture()
    seq, actual = torch.from_numpy(z["aatype"]), torch.from_numpy(z["actual_mask"])
    tb = pocket._tables("cpu")
    n = seq.shape[0]
    for rp in ([0, n], [0, 17, 40, n]):
        rpt = torch.tensor(rp)
        te, cm = pocket.torsion_edges(seq, actual, rpt, tb)
        ft = pocket.node_features(seq, actual, tb)
        for p in range(len(rp) - 1):
            s = slice(rp[p], rp[p + 1])
            te_o, cm_o = opk.build_torsion_edges(seq[s], actual[s], T)
            assert torch.equal(te[s], te_o) and torch.equal(cm[s], cm_o), (rp, p)
            assert torch.equal(ft[s], opk.pocket_features(seq[s], actual[s], T))


@pytest.mark.gpu
def test_gpu_extract_templates_matches_reference_fixture():
    dev = torch.device("cuda:0")
    z = fixture()
    out = pocket.extract_templates(torch.from_numpy(z["aatype"]).to(dev), torch.from_numpy(z["atom14_position"]).float().to(dev))
    torch.cuda.synchronize()
    for k in ("backbone_transl", "backbone_rots", "default_frame", "rigid_group_positions"):
        err = np.abs(out[k].cpu().numpy().astype(np.float64) - z["ref_" + k]).max()
        assert err <= TOL, (k, err)
    assert ang_diff(out["torsion_angle"].cpu().numpy(), z["ref_torsion_angle"]).max() <= 2e-4
    # round trip through the reference-pinned side-chain rebuild: templates -> atom14 == the input coordinates
    a14 = geometry.build_atom14(torch.from_numpy(z["aatype"]), out["backbone_transl"].cpu(), out["backbone_rots"].cpu(),
                                out["default_frame"].cpu(), out["rigid_group_positions"].cpu(), out["torsion_angle"].cpu(),
                                torch.from_numpy(T["atom14_to_group"]).long())
    ideal = torch.from_numpy(z["ideal_mask"]).bool()
    assert (a14[ideal].double() - torch.from_numpy(z["atom14_position"])[ideal]).abs().max() <= 5e-4


@pytest.mark.gpu
def test_gpu_pocket_records_to_poses():
    """Raw coordinates of two pockets (one with missing side chains) -> records -> assembled batch -> sampler."""
    import bench
    import diffbindfr_amd as dba
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(17)
    recs_ref, aa, pos, msk, ptr = [], [], [], [], [0]
    for i, n_atoms in enumerate((90, 70)):
        pk = synthetic.make_pocket(rng, n_atoms)
        rec = synthetic.make_record(pk, synthetic.make_ligand(rng, 12 + 4 * i), rng, drop_sidechains=2 * i)
        a14 = geometry.build_atom14(rec["sequence"], rec["backbone_transl"], rec["backbone_rots"], rec["default_frame"],
                                    rec["rigid_group_positions"], rec["torsion_angle"], torch.from_numpy(T["atom14_to_group"]).long())
        ideal = torch.from_numpy(T["atom14_mask"][rec["sequence"].numpy()]).bool()
        recs_ref.append(rec)
        aa.append(rec["sequence"]); pos.append(a14 * ideal[..., None] + 7.0); msk.append(rec["atom14_mask"])
        ptr.append(ptr[-1] + rec["sequence"].shape[0])
    pockets = pocket.pocket_records(torch.cat(aa), torch.cat(pos), torch.cat(msk), ptr, dev)
    assert len(pockets) == 2
    crs = []
    for pk_rec, rec in zip(pockets, recs_ref):
        # same pocket as the synthetic record up to the rigid shift: masks, edges and features must be identical
        te_o, cm_o = opk.build_torsion_edges(rec["sequence"], rec["atom14_mask"], T)
        assert torch.equal(pk_rec["sc_torsion_edge_mask"], cm_o) and torch.equal(cm_o, rec["sc_torsion_edge_mask"].bool())
        assert torch.equal(pk_rec["torsion_edge_index"], te_o)
        assert torch.equal(pk_rec["pocket_node_feature"], rec["pocket_node_feature"])
        ca = pk_rec["atom14_position"][:, 1]
        assert ca.mean(0).abs().max() < 1e-4                                  # Decentration
        full = {**pk_rec, **{k: rec[k] for k in ("lig_pos", "lig_edge_index", "lig_node", "lig_edge_feat", "tor_edge_mask",
                                                  "rot_node_mask")}}
        crs.append(assemble.ComplexRecord(full))
    samp = dba.DiffBindFRHIP(diffusion_model=bench.seeded_params().to(dev), test_cfg={})
    res = samp.sample_complexes(crs, 2, dev, seed=1)
    assert len(res) == 4 and all(torch.isfinite(l).all() and torch.isfinite(a).all() for l, a in res)


# ------------------------------------------------------------------------------------------------ residue selection

def select_fixture():
    return np.load(os.path.join(GOLDEN, "export.npz")), np.load(os.path.join(GOLDEN, "pocket_select.npz"))


SELECT_CASES = [("any12", 12.0, None, None), ("any8", 8.0, None, None), ("any12_top40", 12.0, 40, None), ("any3", 3.0, None, None),
                ("far", 0.5, None, None), ("ca10", 10.0, None, (1,)), ("bb9_top25", 9.0, 25, (0, 1, 2, 4))]


def test_oracle_pocket_selection_matches_reference_fixture():
    z, ref = select_fixture()
    pos, msk, lig = torch.from_numpy(z["atom37_pos"]), torch.from_numpy(z["atom37_mask"]), torch.from_numpy(z["lig_pos"])
    for name, cut, cap, atoms in SELECT_CASES:
        cols = slice(None) if atoms is None else list(atoms)
        mask, d2 = opk.select_bs(lig, pos[:, cols], msk[:, cols], cut, cap)
        assert np.array_equal(mask.numpy(), ref["ref_" + name]), name
        assert np.abs(d2.numpy() - ref["d2_" + name]).max() <= 1e-4 * max(1.0, float(ref["d2_" + name][ref["d2_" + name] < 1e19].max()))
    assert int(ref["ref_any12"].sum()) == 105 and np.array_equal(ref["ref_any12"], z["pocket_mask"]) and int(ref["ref_far"].sum()) == 1
    mask, _ = opk.select_bs(lig, torch.from_numpy(ref["centroids"])[:, None], msk.bool().any(-1, keepdim=True), 9.0, None)
    assert np.array_equal(mask.numpy(), ref["ref_centroid9"])


def test_select_pocket_has_no_cpu_path():
    z, _ = select_fixture()
    with pytest.raises(Exception):
        pocket.select_pocket(torch.from_numpy(z["atom37_pos"]), z["atom37_mask"], z["lig_pos"])


@pytest.mark.gpu
def test_gpu_select_pocket_matches_reference_fixture():
    z, ref = select_fixture()
    dev = torch.device("cuda:0")
    pos, msk, lig = torch.from_numpy(z["atom37_pos"]).to(dev), torch.from_numpy(z["atom37_mask"]).to(dev), torch.from_numpy(z["lig_pos"]).to(dev)
    for name, cut, cap, atoms in SELECT_CASES:
        mask, d2 = pocket.select_pocket(pos, msk, lig, cut, cap, atoms_id=atoms)
        assert np.array_equal(mask.cpu().numpy(), ref["ref_" + name]), name
        assert np.array_equal(d2.cpu().numpy(), ref["d2_" + name]), name            # same fp32 operation order: bit-equal distances
    mask, _ = pocket.select_pocket(torch.from_numpy(ref["centroids"]).to(dev), msk, lig, 9.0)
    assert np.array_equal(mask.cpu().numpy(), ref["ref_centroid9"])


@pytest.mark.gpu
def test_gpu_select_pocket_many_proteins_vs_oracle():
    """A target-fishing style batch: 37 ragged proteins (1..400 residues, one with no atoms at all in a residue), each with its
    own reference points, one launch; every protein must equal the oracle run on it alone."""
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    sizes = [1, 2, 400] + [int(rng.integers(3, 300)) for _ in range(34)]
    nref = [int(rng.integers(1, 60)) for _ in sizes]
    pos = [(rng.standard_normal((n, 14, 3)) * 2 + rng.standard_normal((n, 1, 3)) * 15).astype(np.float32) for n in sizes]
    msk = [(rng.random((n, 14)) > 0.2).astype(np.float32) for n in sizes]
    msk[2][5] = 0.0
    refs = [(rng.standard_normal((k, 3)) * 8).astype(np.float32) for k in nref]
    res_ptr, ref_ptr = np.concatenate([[0], np.cumsum(sizes)]), np.concatenate([[0], np.cumsum(nref)])
    for cut, cap in ((12.0, None), (6.5, 16), (0.01, None)):
        mask, d2 = pocket.select_pocket(torch.from_numpy(np.concatenate(pos)).to(dev), torch.from_numpy(np.concatenate(msk)).to(dev),
                                        torch.from_numpy(np.concatenate(refs)).to(dev), cut, cap, res_ptr=res_ptr, ref_ptr=ref_ptr)
        mask, d2 = mask.cpu().numpy(), d2.cpu().numpy()
        for p_, n in enumerate(sizes):
            want, wd = opk.select_bs(torch.from_numpy(refs[p_]), torch.from_numpy(pos[p_]), torch.from_numpy(msk[p_]), cut, cap)
            sl = slice(res_ptr[p_], res_ptr[p_ + 1])
            assert np.array_equal(mask[sl], want.numpy()), (cut, cap, p_)
            assert np.allclose(d2[sl], wd.numpy(), rtol=1e-6, atol=0)
            assert mask[sl].sum() >= 1                                       # the nearest residue is always kept


@pytest.mark.gpu
def test_gpu_protein_to_pocket_records_chain():
    """LoadProtein's arrays -> selection -> atom14 -> templates / masks / features in one call, two proteins at once; the 3DBS
    pocket it finds is the fixture's, its templates the reference's (real_3dbs.npz was frozen from the same PDB file)."""
    z, ref = select_fixture()
    real = np.load(os.path.join(GOLDEN, "real_3dbs.npz"))
    dev = torch.device("cuda:0")
    aa = z["aatype"].copy()
    aa[3] = 0 if aa[3] == 20 else aa[3]
    n = aa.shape[0]
    pos = torch.from_numpy(np.concatenate([z["atom37_pos"], z["atom37_pos"] + 100.0])).to(dev)
    msk = torch.from_numpy(np.concatenate([z["atom37_mask"], z["atom37_mask"]])).to(dev)
    lig = torch.from_numpy(np.concatenate([z["lig_pos"], z["lig_pos"][:10] + 100.0])).to(dev)
    recs, mask = pocket.pockets_from_proteins(np.concatenate([aa, aa]), pos, msk, lig, 12.0, res_ptr=[0, n, 2 * n],
                                              ref_ptr=[0, 35, 45])
    assert np.array_equal(mask[:n].cpu().numpy(), z["pocket_mask"]) and len(recs) == 2
    assert 0 < int(mask[n:].sum()) < int(mask[:n].sum())                   # fewer reference atoms -> smaller pocket
    r0 = recs[0]
    same = (r0["sequence"].numpy() == real["aatype"])
    assert r0["sequence"].shape[0] == 105 and same.sum() >= 104
    d = ang_diff(r0["torsion_angle"].numpy(), real["ref_torsion_angle"]) * same[:, None]
    assert d.max() < 2e-4
    assert np.abs((r0["rigid_group_positions"].numpy() - real["ref_rigid_group_positions"]) * same[:, None, None]).max() < TOL
    m14 = np.asarray(real["atom14_mask"]).astype(bool)
    keep_rows = same & (z["atom37_mask"][z["pocket_mask"]][:, 36] == 0)     # rows the export fixture did not edit
    assert np.array_equal(r0["atom14_mask"].numpy()[keep_rows], m14[keep_rows])
    assert abs(float(r0["atom14_position"][r0["atom14_mask"]][:, 0].mean())) < 3.0   # decentred
