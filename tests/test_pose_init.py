"""pose initialisation + batch assembly.

CPU tests: the oracle (oracle/pose_init.py) against the fixture frozen from the reference's own
struct_init.py / druglib.data (tests/golden/pose_init.npz), and the product's host logic
(diffbindfr_amd/assemble.py) against the oracle's collate.
GPU tests: ``dbfr_init_poses`` against the reference fixture and the oracle, and the assembled batch
through the sampler.
"""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import pose_init as opi
from diffbindfr_amd import assemble, lib as L, synthetic
from diffbindfr_amd.packing import PackedBatch
from tests.helpers import GOLDEN

T = synthetic.residue_tables()
Tt = {k: torch.from_numpy(np.asarray(v)) if k == "atom14_to_group" else v for k, v in T.items()}
CASES = ("plain", "fixer", "rigid")
OUT_KEYS = ("lig_pos", "torsion_angle", "rec_atm_pos", "pocket_node_feature", "sc_torsion_edge_mask", "atom14_mask",
            "default_frame", "rigid_group_positions")
POS_TOL = 2e-4     # Angstrom; coordinates reach ~40 A after the sigma = 10 A translation (fp32 ulp there = 4e-6)


def fixture():
    return np.load(os.path.join(GOLDEN, "pose_init.npz"))


def record(z, name):
    pre = name + "_rec_"
    return {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}


def tape(z, pre):
    out = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    out["tr"] = torch.from_numpy(out["tr"])
    return out


def collate_poses(z):
    poses, i = [], 0
    for name in CASES[:2]:
        fixed = opi.sc_fixer(record(z, name), T)
        for _ in range(2):
            poses.append(opi.init_pose(fixed, tape(z, f"collate_tape{i}_"), Tt))
            i += 1
    return poses


@pytest.mark.parametrize("name", CASES)
def test_oracle_pose_init_matches_reference_fixture(name):
    z = fixture()
    rec = record(z, name)
    for seed in (3, 4):
        pre = f"{name}_s{seed}_"
        mine = opi.init_pose(opi.sc_fixer(copy.deepcopy(rec), T), tape(z, pre + "tape_"), Tt)
        for k in OUT_KEYS:
            ref = torch.from_numpy(z[pre + "out_" + k])
            assert mine[k].shape == ref.shape, k
            assert (mine[k].double() - ref.double()).abs().max() <= 1e-6, (name, seed, k)
    if name == "fixer":      # the case must really exercise SCFixer
        assert (z["fixer_rec_atom14_mask"] != z["fixer_s3_out_atom14_mask"]).sum() > 0


def test_oracle_collate_matches_reference_fixture():
    z = fixture()
    mine = opi.collate(collate_poses(z))
    keys = [k[len("collate_"):] for k in z.files if k.startswith("collate_") and "tape" not in k]
    assert len(keys) >= 20
    for k in keys:
        ref = torch.from_numpy(z["collate_" + k])
        assert mine[k].dtype == ref.dtype and mine[k].shape == ref.shape, k
        assert (mine[k].double() - ref.double()).abs().max() <= 1e-6, k


def test_product_sc_fixer_matches_oracle():
    z = fixture()
    for name in CASES:
        rec = record(z, name)
        fixed = opi.sc_fixer(copy.deepcopy(rec), T)
        cr = assemble.ComplexRecord(copy.deepcopy(rec))
        assert torch.equal(cr.sc_mask, fixed["sc_torsion_edge_mask"].bool())
        assert torch.equal(cr.atom14_mask, fixed["atom14_mask"].bool())
        assert torch.equal(cr.default_frame, fixed["default_frame"])
        assert torch.equal(cr.rigid_group_positions, fixed["rigid_group_positions"])
    # backbone atom missing anywhere in the pocket: the reference zeroes every chi mask (struct_init.py:83-92)
    rec = record(z, "fixer")
    rec["atom14_mask"][0, 1] = False
    fixed = opi.sc_fixer(copy.deepcopy(rec), T)
    cr = assemble.ComplexRecord(copy.deepcopy(rec))
    assert int(cr.sc_mask.sum()) == 0 == int(fixed["sc_torsion_edge_mask"].sum())
    assert torch.equal(cr.atom14_mask, fixed["atom14_mask"].bool())


def _records_and_reference_batch(poses_per_complex, seed=5):
    """Synthetic records -> (ComplexRecords, oracle collate of oracle-initialised poses, the tapes used)."""
    rng = np.random.default_rng(seed)
    recs = [synthetic.make_record(synthetic.make_pocket(rng, 70), synthetic.make_ligand(rng, 16), rng, drop_sidechains=2),
            synthetic.make_record(synthetic.make_pocket(rng, 50), synthetic.make_ligand(rng, 5), rng),
            synthetic.make_record(synthetic.make_pocket(rng, 90), synthetic.make_ligand(rng, 24), rng)]
    recs[1]["tor_edge_mask"].zero_()                       # a rigid ligand in the middle of the batch
    recs[1]["rot_node_mask"] = recs[1]["rot_node_mask"][:0]
    poses, tapes = [], []
    for rec, n in zip(recs, poses_per_complex):
        fixed = opi.sc_fixer(copy.deepcopy(rec), T)
        for _ in range(n):
            n_tor = int(rec["tor_edge_mask"].sum())
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            x, y, zz, w = q
            R = np.array([[1 - 2 * (y * y + zz * zz), 2 * (x * y - zz * w), 2 * (x * zz + y * w)],
                          [2 * (x * y + zz * w), 1 - 2 * (x * x + zz * zz), 2 * (y * zz - x * w)],
                          [2 * (x * zz - y * w), 2 * (y * zz + x * w), 1 - 2 * (x * x + y * y)]])
            tp = dict(tor=rng.uniform(-np.pi, np.pi, n_tor).astype(np.float32).astype(np.float64), rot=R.astype(np.float32).astype(np.float64),
                      tr=torch.from_numpy(rng.normal(0, 10, (1, 3))).float(),
                      sc=rng.uniform(-np.pi, np.pi, (rec["sequence"].shape[0], 4)).astype(np.float32).astype(np.float64))
            tapes.append(tp)
            poses.append(opi.init_pose(fixed, tp, Tt))
    return recs, opi.collate(poses), tapes


def _batch_tape(tapes, dev):
    tor = np.concatenate([t["tor"] for t in tapes]) if any(len(t["tor"]) for t in tapes) else np.zeros(1)
    return dict(tor=torch.from_numpy(tor).float().to(dev), rot=torch.from_numpy(np.stack([t["rot"] for t in tapes])).float().to(dev),
                tr=torch.cat([t["tr"] for t in tapes]).to(dev), sc=torch.from_numpy(np.concatenate([t["sc"] for t in tapes])).float().to(dev))


def test_assemble_matches_collate_then_pack():
    """assemble(records, poses) must produce the very tensors PackedBatch makes from the reference-format collated
    batch (every static field bit for bit; lig_pos = tiled input conformers; psi kept)."""
    n_poses = [3, 2, 2]
    recs, coll, _ = _records_and_reference_batch(n_poses)
    ref = PackedBatch(coll, "cpu")
    crs = [assemble.ComplexRecord(copy.deepcopy(r)) for r in recs]
    pb = assemble.assemble(crs, n_poses, "cpu")
    assert pb.dims == ref.dims
    for k in L._BATCH_PTRS:
        if k in ("lig_pos", "rec_pos", "torsion_angle"):
            continue
        assert pb.t[k].dtype == ref.t[k].dtype and pb.t[k].shape == ref.t[k].shape, k
        assert torch.equal(pb.t[k], ref.t[k]), k
    assert torch.equal(pb.t["torsion_angle"][:, 0], ref.t["torsion_angle"][:, 0])
    lp = pb.lig_ptr_host.tolist()
    g = 0
    for r, n in zip(recs, n_poses):
        for _ in range(n):
            assert torch.equal(pb.t["lig_pos"][lp[g]:lp[g + 1]], r["lig_pos"])
            g += 1
    assert torch.equal(pb.sc_mask, ref.sc_mask) and torch.equal(pb.atom14_mask, ref.atom14_mask)
    # an int for `poses` = the same count for every complex
    pb2 = assemble.assemble(crs, 2, "cpu")
    assert pb2.G == 6 and pb2.dims["NL"] == 2 * sum(r["lig_pos"].shape[0] for r in recs)


# ------------------------------------------------------------------------------------------------ GPU
def _model(dev):
    import bench
    return bench.seeded_params().to(dev)


@pytest.mark.gpu
def test_gpu_init_poses_matches_reference_fixture():
    """dbfr_init_poses on the fixture's records with the fixture's draws vs the REFERENCE's outputs."""
    dev = torch.device("cuda:0")
    model = _model(dev)
    z = fixture()
    for name in CASES:
        rec = record(z, name)
        for seed in (3, 4):
            pre = f"{name}_s{seed}_"
            tp = tape(z, pre + "tape_")
            pb = assemble.assemble([assemble.ComplexRecord(copy.deepcopy(rec))], 1, dev)
            bt = _batch_tape([dict(tor=tp.get("tor", np.zeros(0)), rot=tp["rot"], tr=tp["tr"], sc=tp["sc"])], dev)
            a14 = torch.zeros(pb.dims["NR"], 14, 3, device=dev)
            assemble.init_poses(model, pb, bt, a14)
            torch.cuda.synchronize()
            for k, mine in (("lig_pos", pb.lig_pos), ("rec_atm_pos", pb.rec_pos), ("torsion_angle", pb.torsion_angle)):
                ref = torch.from_numpy(z[pre + "out_" + k])
                err = (mine.cpu().double() - ref.double()).abs().max().item()
                assert err <= POS_TOL, (name, seed, k, err)
            m14 = torch.from_numpy(z[pre + "out_atom14_mask"]).bool()
            assert (a14.cpu()[m14] - torch.from_numpy(z[pre + "out_rec_atm_pos"])).abs().max() <= POS_TOL
            assert float(a14.cpu()[~m14].abs().max()) == 0.0


@pytest.mark.gpu
def test_gpu_assemble_init_matches_oracle_and_samples():
    """7 poses of 3 complexes: device initialisation vs the oracle pose by pose, then the assembled batch and the
    collate->PackedBatch batch must give bit-identical sampler output from the same state."""
    import diffbindfr_amd as dba
    dev = torch.device("cuda:0")
    model = _model(dev)
    n_poses = [3, 2, 2]
    recs, coll, tapes = _records_and_reference_batch(n_poses)
    crs = [assemble.ComplexRecord(copy.deepcopy(r)) for r in recs]
    pb = assemble.assemble(crs, n_poses, dev)
    assemble.init_poses(model, pb, _batch_tape(tapes, dev))
    torch.cuda.synchronize()
    for k, ref in (("lig_pos", coll["lig_pos"]), ("rec_pos", coll["rec_atm_pos"]), ("torsion_angle", coll["torsion_angle"])):
        err = (pb.t[k].cpu().double() - ref.double()).abs().max().item()
        assert err <= POS_TOL, (k, err)
    ref_pb = PackedBatch({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in coll.items()}, dev)
    for k in ("lig_pos", "rec_pos", "torsion_angle"):
        ref_pb.t[k].copy_(pb.t[k])
    samp = dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})
    recs_s, _ = samp.schedule()
    Tn = len(recs_s)
    gen = torch.Generator(device=dev).manual_seed(3)
    zt = {"tr": torch.randn(Tn, pb.G, 3, device=dev, generator=gen), "rot": torch.randn(Tn, pb.G, 3, device=dev, generator=gen),
          "tor": torch.randn(Tn, max(pb.dims["NTOR"], 1), device=dev, generator=gen),
          "sc": torch.randn(Tn, max(pb.dims["NSC"], 1), device=dev, generator=gen)}
    la, aa = samp.sample_packed(pb, zt)
    la, aa = la.clone(), aa.clone()
    lb, ab = samp.sample_packed(ref_pb, zt)
    assert torch.isfinite(la).all() and torch.equal(la, lb) and torch.equal(aa, ab)


@pytest.mark.gpu
def test_gpu_draw_init_tape_statistics():
    """The device tape: proper rotations, N(0, 10) translations, U(-pi, pi) angles; initialised poses keep bond lengths."""
    dev = torch.device("cuda:0")
    model = _model(dev)
    rng = np.random.default_rng(11)
    rec = synthetic.make_record(synthetic.make_pocket(rng, 60), synthetic.make_ligand(rng, 20), rng)
    cr = assemble.ComplexRecord(rec)
    pb = assemble.assemble([cr], 512, dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    tp = assemble.draw_init_tape(pb, generator=gen)
    R = tp["rot"]
    eye = torch.eye(3, device=dev).expand_as(R)
    assert (R @ R.transpose(1, 2) - eye).abs().max() < 1e-5 and (torch.linalg.det(R) - 1).abs().max() < 1e-5
    assert abs(float(tp["tr"].std()) - 10.0) < 0.6 and abs(float(tp["tr"].mean())) < 1.0
    for k in ("tor", "sc"):
        assert float(tp[k].min()) >= -np.pi and float(tp[k].max()) <= np.pi and abs(float(tp[k].mean())) < 0.1
    assemble.init_poses(model, pb, tp)
    torch.cuda.synchronize()
    pos = pb.lig_pos.reshape(512, cr.n_l, 3)
    ref = cr.lig_pos.to(dev)
    bs, bd = cr.bond_src.to(dev), cr.bond_dst.to(dev)
    bl = lambda p: (p[..., bs, :] - p[..., bd, :]).norm(dim=-1)
    assert (bl(pos) - bl(ref)).abs().max() < 1e-4          # torsion kicks + rigid motion preserve every bond length
    cen = pos.mean(1)
    assert (cen - tp["tr"]).abs().max() < 1e-4             # the centroid lands on the drawn translation
    assert torch.isfinite(pb.rec_pos).all() and float(pb.rec_pos.abs().max()) > 0


@pytest.mark.gpu
def test_gpu_sample_complexes_end_to_end():
    """records -> poses in one call; same seed => identical poses, different seed => different poses."""
    import diffbindfr_amd as dba
    dev = torch.device("cuda:0")
    samp = dba.DiffBindFRHIP(diffusion_model=_model(dev), test_cfg={})
    rng = np.random.default_rng(21)
    recs = [synthetic.make_record(synthetic.make_pocket(rng, 80), synthetic.make_ligand(rng, 18), rng) for _ in range(2)]
    a = samp.sample_complexes(recs, [3, 2], dev, seed=9)
    b = samp.sample_complexes(recs, [3, 2], dev, seed=9)
    c = samp.sample_complexes(recs, [3, 2], dev, seed=10)
    assert len(a) == 5
    for g, (lig, a14) in enumerate(a):
        r = recs[0] if g < 3 else recs[1]
        assert lig.shape == (1, r["lig_pos"].shape[0], 3) and a14.shape == (1, r["sequence"].shape[0], 14, 3)
        assert torch.isfinite(lig).all() and torch.isfinite(a14).all()
        assert torch.equal(lig, b[g][0]) and torch.equal(a14, b[g][1])
        assert not torch.equal(lig, c[g][0])
    # poses of one complex differ from each other (independent initialisation + noise)
    assert not torch.equal(a[0][0], a[1][0])
