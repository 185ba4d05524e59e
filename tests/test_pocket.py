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
