"""the output side of the sampler -- per-pose metrics and PDB text.

CPU: the oracle (oracle/export.py) against the fixture frozen from the reference's own calc_lig_centroid /
sidechain_rmsd / symm_rmsd / Protein.pos_update + to_pdb (tests/golden/export.npz, 3DBS example); the chi part, whose
dihedral extraction the reference delegates to its vendored openfold transforms (pinned by chi_differ.npz), cross-checked against the chi angles of
the reference-pinned extract_chi_and_template; the library's host-side PDB writer byte-for-byte against the reference
text; the product's automorphism search against the reference's networkx matcher.
GPU: dbfr_pose_metrics against the reference fixture and the oracle.
"""
import os

import numpy as np
import pytest
import torch

from oracle import export as oex, geometry, pocket as opk
from diffbindfr_amd import export as pex, ligand, synthetic
from diffbindfr_amd.lib import DbfrError
from tests.helpers import GOLDEN

T = synthetic.residue_tables()
TOL = 2e-5        # Angstrom, relative to coordinates of up to ~80 A in fp32 (reference and device both fp32)


def fixture():
    return np.load(os.path.join(GOLDEN, "export.npz"))


def topology(z, remark=True):
    return pex.ProteinTopology(z["aatype"], z["atom37_pos"], z["atom37_mask"], z["residue_index"], z["chain_index"], z["b_factors"],
                               str(z["remark"]) if remark else None, np.nonzero(z["pocket_mask"])[0])


def absolute(z):
    c = torch.from_numpy(z["center"])
    return oex.add_center_pos(torch.from_numpy(z["lig_traj"]), c), oex.add_center_pos(torch.from_numpy(z["prot_traj"]), c), c


def test_oracle_metrics_match_reference_fixture():
    z = fixture()
    lt, pt, c = absolute(z)
    seq, tm = torch.from_numpy(z["aatype"][z["pocket_mask"]]), torch.from_numpy(z["target_atom14_mask"]).float()
    cen = oex.calc_lig_centroid(lt, torch.from_numpy(z["lig_pos"]))
    assert np.abs(cen.numpy() - z["ref_centroid"]).max() <= 1e-6
    sc = oex.sidechain_rmsd(pt, oex.add_center_pos(torch.from_numpy(z["target_atom14"]), c), tm, seq, T)
    assert np.abs(sc.numpy() - z["ref_sc_rmsd"]).max() <= 1e-6
    assert z["ref_sc_rmsd"][0, 0] < 1e-5 and z["ref_sc_rmsd"][1, 2] < 1e-5      # identical pose / the other atom naming
    assert z["ref_sc_rmsd"][3, 0] > 1.0
    perms = [(p, np.arange(p.shape[0])) for p in z["ref_perms"]]
    rm = oex.symm_rmsd(perms, z["ha_mask"], z["lig_pos"], lt.numpy())
    assert np.abs(rm.numpy() - z["ref_symm_rmsd"]).max() <= 1e-6
    ident = oex.symm_rmsd(perms[:1] if (z["ref_perms"][0] == np.arange(35)).all() else [(np.arange(35), np.arange(35))],
                          z["ha_mask"], z["lig_pos"], lt.numpy())
    assert (rm <= ident + 1e-7).all()


def test_oracle_chi_matches_reference_pinned_chi_extraction():
    """openfold's atom37_to_torsion_angles restated in oracle/export.chi_sin_cos (pinned by chi_differ.npz): its angles must be the
    chi angles the reference's own extract_chi_and_template recovers from the same coordinates, and the difference of
    two structures built from known torsions must be the applied rotation (wrapped; pi-periodic chis modulo pi)."""
    z = fixture()
    seq = z["aatype"][z["pocket_mask"]]
    tgt = z["target_atom14"].astype(np.float64)
    msk = z["target_atom14_mask"]
    keep = seq != 20
    seq, tgt, msk = seq[keep], tgt[keep], msk[keep]
    ref = opk.extract_chi_and_template(seq, tgt.copy(), T["atom14_mask"][seq][..., None].astype(np.float32), T)
    sc, alt, m = oex.chi_sin_cos(torch.from_numpy(tgt).float(), torch.from_numpy(msk).float(), torch.from_numpy(seq), T)
    ang = torch.atan2(sc[..., 0], sc[..., 1]).numpy()
    d = np.abs(np.angle(np.exp(1j * (ang - ref["torsion_angle"][:, 1:]))))
    assert (d * m.numpy()).max() < 2e-4 and m.sum() > 150
    # known rotations: rebuild the pocket from its templates with chi + delta (geometry.build_atom14 is pinned against the
    # reference's build_pdb_from_template)
    rng = np.random.default_rng(3)
    delta = rng.uniform(-np.pi, np.pi, (seq.shape[0], 4)).astype(np.float32) * m.numpy()
    tor = torch.from_numpy(ref["torsion_angle"].astype(np.float32))
    build = lambda t: geometry.build_atom14(torch.from_numpy(seq), torch.from_numpy(ref["backbone_transl"]).float(),
                                            torch.from_numpy(ref["backbone_rots"]).float(), torch.from_numpy(ref["default_frame"]),
                                            torch.from_numpy(ref["rigid_group_positions"]), t, torch.from_numpy(T["atom14_to_group"]))
    base = build(tor)
    tor2 = tor.clone()
    tor2[:, 1:] += torch.from_numpy(delta)
    moved = build(tor2)
    mt = torch.from_numpy(msk).float()
    got, mask = oex.chi_differ(moved[None], base, mt, torch.from_numpy(seq), T)
    want = np.abs(np.angle(np.exp(1j * delta.astype(np.float64))))
    period = T["chi_pi_periodic"][seq]
    want = np.where(period > 0, np.minimum(want, np.pi - want), want) * m.numpy()
    # the reference's fmod wrap turns differences below -pi into pi: leave those entries out of the comparison
    t_ang = torch.atan2(*oex.chi_sin_cos(base, mt, torch.from_numpy(seq), T)[0].unbind(-1)).numpy()
    p_ang = torch.atan2(*oex.chi_sin_cos(moved, mt, torch.from_numpy(seq), T)[0].unbind(-1)).numpy()
    t_alt = np.where(t_ang > 0, t_ang - np.pi, t_ang + np.pi)                     # atan2(-sin, -cos)
    regular = (t_ang - p_ang + np.pi >= 1e-3) & ((period == 0) | (t_alt - p_ang + np.pi >= 1e-3))
    assert regular.sum() > 100
    assert (np.abs(got[0].numpy() - want) * regular).max() < 5e-4
    rate = oex.chi_success_rate(got, mask)
    assert rate.shape == (1, 4) and ((rate >= 0) & (rate <= 1)).all()


def test_oracle_pdb_text_matches_reference_fixture():
    z = fixture()
    _, pt, _ = absolute(z)
    rows = np.nonzero(z["pocket_mask"])[0]
    for pid in (0, 3):
        txt = oex.pose_pdb(z["aatype"], z["atom37_pos"], z["atom37_mask"], z["residue_index"], z["chain_index"], z["b_factors"], rows,
                           pt[pid, -1].numpy(), T, str(z["remark"]))
        assert txt == bytes(z[f"ref_pdb_full_{pid}"]).decode()
    txt = oex.to_pdb(z["strip_aatype"], z["strip_pos"], z["strip_mask"], z["strip_resid"], z["strip_chain"], z["strip_bfac"], T, None)
    assert txt == bytes(z["ref_pdb_strip"]).decode()


def test_library_pdb_writer_is_byte_exact(tmp_path):
    """dbfr_pdb_format / dbfr_pdb_write_files (host code of the C ABI) against the text the reference's to_pdb produced:
    protein with the pocket residues replaced by a pose (UNK residue, OXT atoms, two chains 'A' / 'BA'), the pocket alone,
    and a synthetic strip (30 chains, negative and 4-digit residue numbers, coordinates wider than the 8-column field)."""
    z = fixture()
    _, pt, _ = absolute(z)
    topo = topology(z)
    for pid in (0, 3):
        assert topo.to_pdb(pt[pid, -1]) == bytes(z[f"ref_pdb_full_{pid}"]).decode()
        assert topo.pocket().to_pdb(pt[pid, -1].numpy()) == bytes(z[f"ref_pdb_pkt_{pid}"]).decode()
    strip = pex.ProteinTopology(z["strip_aatype"], z["strip_pos"], z["strip_mask"], z["strip_resid"], z["strip_chain"], z["strip_bfac"])
    ref = bytes(z["ref_pdb_strip"]).decode()
    txt = strip.to_pdb(model=2)                          # model > 1: no dated REMARK line (protein.py:709-711)
    assert txt == ref[:-81] + "ENDMDL".ljust(80) + "\n" + ref[-81:]
    dated = strip.to_pdb()
    assert dated.startswith("REMARK   1 CREATED WITH MDLDruglib 1.0.0, ") and dated.split("\n", 1)[1] == ref
    # files: every sample's prot_final.pdb in one call, written by library threads
    paths = [str(tmp_path / f"sample_{i + 1}" / "prot_final.pdb") for i in range(4)]
    for p in paths:
        os.makedirs(os.path.dirname(p))
    topo.write_poses(pt[:, -1].numpy(), paths, threads=3)
    for pid in (0, 3):
        assert open(paths[pid]).read() == bytes(z[f"ref_pdb_full_{pid}"]).decode()
    assert open(paths[1]).read() == topo.to_pdb(pt[1, -1])
    with pytest.raises(DbfrError):
        topo.write_poses(pt[:, -1].numpy(), [str(tmp_path / "missing_dir" / "x.pdb")] * 4)
    with pytest.raises(DbfrError):
        pex.ProteinTopology(np.array([21]), np.zeros((1, 37, 3)), np.ones((1, 37)), np.array([1]), np.array([0]), np.zeros((1, 37))).to_pdb()
    with pytest.raises(DbfrError):
        topo.to_pdb(pt[0, -1], rows=np.array([0, 1]))


def test_library_pdb_writer_equals_oracle_on_random_structures():
    """Empty structure, single residue, and random strips (missing atoms, any chain / residue numbering, poses covering
    all / some / no residues) -- the library text must equal the reference-pinned oracle's."""
    rng = np.random.default_rng(101)
    for case in range(24):
        n = [0, 1][case] if case < 2 else int(rng.integers(2, 40))
        aa = rng.integers(0, 21, n)
        m37 = (T["atom37_mask"][aa] * (rng.random((n, 37)) > 0.15)).astype(np.float32)
        if n:
            m37[rng.integers(0, n), 36] = 1.0                                         # an OXT somewhere
        pos = (rng.standard_normal((n, 37, 3)) * rng.choice([3.0, 60.0, 700.0])).astype(np.float32)
        chain = np.sort(rng.integers(0, 60, n))
        resid = rng.integers(-99, 9999, n)
        bfac = np.round(rng.random((n, 37)) * 150, int(rng.integers(0, 5)))
        remark = None if case % 3 == 0 else "REMARK   1 TEST %d" % case
        topo = pex.ProteinTopology(aa, pos, m37, resid, chain, bfac, remark)
        model = None if case % 2 else 3
        want = oex.to_pdb(aa, pos, m37, resid, chain, bfac, T, remark, model=model, add_end=bool(case % 4))
        got = topo.to_pdb(model=model, add_end=bool(case % 4), version="x")
        if remark is None and model is None:
            got = got.split("\n", 1)[1]                                               # the dated REMARK line
        assert got == want, case
        if n >= 2:
            rows = np.sort(rng.choice(n, int(rng.integers(1, n + 1)), replace=False))
            p14 = (rng.standard_normal((rows.shape[0], 14, 3)) * 30).astype(np.float32)
            want = oex.pose_pdb(aa, pos, m37, resid, chain, bfac, rows, p14, T, "REMARK X")
            topo.remark = "REMARK X"
            assert topo.to_pdb(p14, rows=rows) == want, case


def test_automorphisms_match_the_reference_matcher():
    z = fixture()
    mine = ligand.automorphisms(z["lig_elements"], z["lig_edge_index"])
    assert {tuple(p) for p in mine.tolist()} == {tuple(p) for p in z["ref_perms"].tolist()} and len(mine) == len(z["ref_perms"]) == 8
    assert (mine[0] == np.arange(35)).all() or any((p == np.arange(35)).all() for p in mine)
    nx = pytest.importorskip("networkx")
    from networkx.algorithms.isomorphism import GraphMatcher
    rng = np.random.default_rng(11)
    for i in range(8):
        lg = synthetic.make_ligand(rng, int(rng.integers(3, 26)))
        n, ei = lg["n_lig"], lg["lig_edge_index"]
        lab = rng.integers(0, 2, n)
        elab = None
        if i % 2:
            und = {}
            elab = np.asarray([und.setdefault((min(u, v), max(u, v)), int(rng.integers(0, 2))) for u, v in ei.T.tolist()])
        g = nx.Graph()
        for a in range(n):
            g.add_node(a, aprops=int(lab[a]))
        for k, (u, v) in enumerate(ei.T.tolist()):
            if u < v:
                g.add_edge(u, v, **({} if elab is None else {"eprops": int(elab[k])}))
        gm = GraphMatcher(g, g, lambda a, b: a["aprops"] == b["aprops"], None if elab is None else (lambda a, b: a["eprops"] == b["eprops"]))
        want = set()
        for iso in gm.isomorphisms_iter():
            keys, vals = np.array(list(iso.keys())), np.array(list(iso.values()))
            want.add(tuple(keys[np.argsort(vals)].tolist()))            # match_graphs' sorted_array (isom_graph.py:113-126)
        got = ligand.automorphisms(lab, ei, elab)
        assert {tuple(p) for p in got.tolist()} == want and len(got) == len(want), i
    with pytest.raises(ValueError):
        ligand.automorphisms(np.zeros(9, int), np.array([[0] * 8 + list(range(1, 9)), list(range(1, 9)) + [0] * 8]), limit=100)


def test_pose_metrics_has_no_cpu_path():
    z = fixture()
    with pytest.raises(DbfrError):
        pex.pose_metrics(torch.from_numpy(z["lig_traj"]), torch.from_numpy(z["prot_traj"]), z["center"], z["lig_pos"], z["target_atom14"],
                         z["target_atom14_mask"], z["aatype"][z["pocket_mask"]])


# ------------------------------------------------------------------------------------------------------------ GPU

def oracle_metrics(lig_traj, prot_traj, center, lig_pos, tgt14, tmask, seq, perms, ha):
    c = torch.as_tensor(center).float()
    lt, pt = oex.add_center_pos(lig_traj, c), oex.add_center_pos(prot_traj, c)
    tm = torch.as_tensor(tmask).float()
    delta, mask = oex.chi_differ(pt, torch.as_tensor(tgt14).float(), tm, seq, T)
    return dict(centroid=oex.calc_lig_centroid(lt, torch.as_tensor(lig_pos).float()),
                sc_rmsd=oex.sidechain_rmsd(pt, oex.add_center_pos(torch.as_tensor(tgt14).float(), c), tm, seq, T),
                delta_chi=delta, chi_rate=oex.chi_success_rate(delta, mask),
                lig_rmsd=oex.symm_rmsd([(np.asarray(p), np.arange(len(p))) for p in perms], np.asarray(ha, bool), np.asarray(lig_pos),
                                       lt.numpy()))


@pytest.mark.gpu
def test_gpu_pose_metrics_match_reference_fixture():
    z = fixture()
    dev = torch.device("cuda:0")
    seq = z["aatype"][z["pocket_mask"]]
    perms = ligand.automorphisms(z["lig_elements"], z["lig_edge_index"])
    out = pex.pose_metrics(torch.from_numpy(z["lig_traj"]).to(dev), torch.from_numpy(z["prot_traj"]).to(dev), z["center"], z["lig_pos"],
                           z["target_atom14"], z["target_atom14_mask"], seq, perms=perms, heavy_mask=z["ha_mask"], with_delta_chi=True)
    torch.cuda.synchronize()
    assert np.abs(out["centroid"].cpu().numpy() - z["ref_centroid"]).max() < TOL
    assert np.abs(out["sc_rmsd"].cpu().numpy() - z["ref_sc_rmsd"]).max() < TOL
    assert np.abs(out["lig_rmsd"].cpu().numpy() - z["ref_symm_rmsd"]).max() < TOL
    want = oracle_metrics(torch.from_numpy(z["lig_traj"]), torch.from_numpy(z["prot_traj"]), z["center"], z["lig_pos"], z["target_atom14"],
                          z["target_atom14_mask"], torch.from_numpy(seq), perms, z["ha_mask"])
    d = np.abs(out["delta_chi"].cpu().numpy() - want["delta_chi"].numpy())
    # an angle next to the reference's wrap discontinuity may land on either side in fp32: allow a handful
    assert (d > 1e-3).sum() <= 2 and np.median(d) < 1e-5
    assert np.abs(out["chi_rate"].cpu().numpy() - want["chi_rate"].numpy()).max() < 0.02
    assert out["chi_rate"][0, 0].min().item() == 1.0 and out["delta_chi"][0, 0].abs().max().item() < 1e-3


@pytest.mark.gpu
def test_gpu_pose_metrics_empty_and_single():
    dev = torch.device("cuda:0")
    z = fixture()
    seq = z["aatype"][z["pocket_mask"]]
    args = (z["center"], z["lig_pos"], z["target_atom14"], z["target_atom14_mask"], seq)
    out = pex.pose_metrics(torch.zeros(0, 3, 35, 3, device=dev), torch.zeros(0, 3, seq.shape[0], 14, 3, device=dev), *args)
    assert out["centroid"].shape == (0, 3) and out["chi_rate"].shape == (0, 3, 4)
    one = pex.pose_metrics(torch.from_numpy(z["lig_traj"][2:3, 1:2]).to(dev), torch.from_numpy(z["prot_traj"][2:3, 1:2]).to(dev), *args)
    torch.cuda.synchronize()
    assert abs(one["centroid"].item() - z["ref_centroid"][2, 1]) < TOL and abs(one["sc_rmsd"].item() - z["ref_sc_rmsd"][2, 1]) < TOL
    a = pex.pose_metrics(torch.from_numpy(z["lig_traj"]).to(dev), torch.from_numpy(z["prot_traj"]).to(dev), *args, with_delta_chi=True)
    b = pex.pose_metrics(torch.from_numpy(z["lig_traj"]).to(dev), torch.from_numpy(z["prot_traj"]).to(dev), *args, with_delta_chi=True)
    torch.cuda.synchronize()
    for key in ("centroid", "sc_rmsd", "chi_rate", "lig_rmsd", "delta_chi"):
        assert torch.equal(a[key], b[key]), key                  # fixed-order reductions: bitwise reproducible


@pytest.mark.gpu
def test_gpu_pose_metrics_tiles_and_many_automorphisms():
    """More residues than one LDS tile (128), a ligand longer than a wave, hundreds of automorphisms, no heavy mask."""
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(17)
    n_res = 301
    seq = rng.integers(0, 20, n_res)
    tmask = T["atom14_mask"][seq] * (rng.random((n_res, 14)) > 0.03)
    tgt = (rng.standard_normal((n_res, 14, 3)) * 3 + rng.standard_normal((n_res, 1, 3)) * 12).astype(np.float32) * tmask[..., None]
    n_lig = 150
    lig = (rng.standard_normal((n_lig, 3)) * 6 + 30).astype(np.float32)
    perms = np.stack([np.arange(n_lig)] + [rng.permutation(n_lig) for _ in range(299)]).astype(np.int32)
    P, F = 3, 5
    center = np.array([31.0, -4.5, 12.25], np.float32)
    pt = torch.from_numpy(tgt)[None, None] + 0.7 * torch.randn(P, F, n_res, 14, 3, generator=torch.Generator().manual_seed(1))
    pt = pt * torch.from_numpy(tmask)[None, None, :, :, None].float()
    lt = torch.from_numpy(lig - center)[None, None] + 0.5 * torch.randn(P, F, n_lig, 3, generator=torch.Generator().manual_seed(2))
    lt[1, 1] = torch.from_numpy(lig - center)[torch.from_numpy(perms[77]).long().argsort()]   # pose[perm[a]] = target[a]
    out = pex.pose_metrics(lt.to(dev), pt.float().to(dev), center, lig, tgt, tmask, seq, perms=perms, with_delta_chi=True)
    torch.cuda.synchronize()
    want = oracle_metrics(lt, pt.float(), center, lig, tgt, tmask, torch.from_numpy(seq), perms, np.ones(n_lig, bool))
    for k in ("centroid", "sc_rmsd", "lig_rmsd"):
        assert np.abs(out[k].cpu().numpy() - want[k].numpy()).max() < 5e-5, k
    assert out["lig_rmsd"][1, 1].item() < 1e-4
    d = np.abs(out["delta_chi"].cpu().numpy() - want["delta_chi"].numpy())
    assert (d > 1e-3).sum() <= 4 and np.median(d) < 1e-5
    assert np.abs(out["chi_rate"].cpu().numpy() - want["chi_rate"].numpy()).max() < 0.02
    # a small ligand with more automorphisms than the per-lane path keeps in LDS (falls back to waves over automorphisms)
    n2 = 20
    lig2 = lig[:n2]
    many = np.stack([np.arange(n2)] + [rng.permutation(n2) for _ in range(2999)]).astype(np.int32)
    lt2 = torch.from_numpy(lig2 - center)[None, None] + 0.3 * torch.randn(2, 3, n2, 3, generator=torch.Generator().manual_seed(3))
    hv = rng.random(n2) > 0.2
    got2 = pex.pose_metrics(lt2.to(dev), pt[:2, :3].float().to(dev), center, lig2, tgt, tmask, seq, perms=many, heavy_mask=hv)
    few2 = pex.pose_metrics(lt2.to(dev), pt[:2, :3].float().to(dev), center, lig2, tgt, tmask, seq, perms=many[:50], heavy_mask=hv)
    torch.cuda.synchronize()
    lt2_abs = oex.add_center_pos(lt2, torch.from_numpy(center)).numpy()
    want2 = oex.symm_rmsd([(p_, np.arange(n2)) for p_ in many], hv, lig2, lt2_abs)
    assert np.abs(got2["lig_rmsd"].cpu().numpy() - want2.numpy()).max() < 5e-5
    want3 = oex.symm_rmsd([(p_, np.arange(n2)) for p_ in many[:50]], hv, lig2, lt2_abs)
    assert np.abs(few2["lig_rmsd"].cpu().numpy() - want3.numpy()).max() < 5e-5
    assert np.abs(few2["centroid"].cpu().numpy() - got2["centroid"].cpu().numpy()).max() < 5e-5
    # only what is asked for is computed: ligand-only call
    lib_only = pex.pose_metrics(lt.to(dev), pt.float().to(dev), center, lig, tgt, tmask, seq)
    torch.cuda.synchronize()
    ident = oex.symm_rmsd([(np.arange(n_lig), np.arange(n_lig))], np.ones(n_lig, bool), lig, oex.add_center_pos(lt, torch.from_numpy(center)).numpy())
    assert np.abs(lib_only["lig_rmsd"].cpu().numpy() - ident.numpy()).max() < 5e-5


@pytest.mark.gpu
def test_gpu_metrics_of_sampled_poses_stay_on_the_device():
    """End of the pipeline: trajectories leave the sampler in HBM and go straight into the metrics launch; same numbers
    as the oracle computes from host copies.  Target = the record's own (crystal) structure."""
    import bench
    import diffbindfr_amd as dba
    from diffbindfr_amd import assemble
    dev = torch.device("cuda:0")
    samp = dba.DiffBindFRHIP(diffusion_model=bench.seeded_params().to(dev), test_cfg={})
    rng = np.random.default_rng(23)
    recs = [synthetic.make_record(synthetic.make_pocket(rng, 70), synthetic.make_ligand(rng, 16), rng) for _ in range(2)]
    poses = [3, 2]
    res = samp.sample_complexes(recs, poses, dev, seed=4, visualize=True, keep_on_device=True)
    g = 0
    for rec, n in zip(recs, poses):
        cr = assemble.ComplexRecord(rec)
        lig_traj = torch.stack([res[g + i][0] for i in range(n)])           # [P,T,N_l,3]
        prot_traj = torch.stack([res[g + i][1] for i in range(n)])
        g += n
        assert lig_traj.is_cuda and prot_traj.is_cuda and lig_traj.shape[1] == 20
        seq, tmask = cr.sequence, cr.atom14_mask.float()
        tgt = geometry.build_atom14(seq, cr.backbone_transl, cr.backbone_rots, cr.default_frame, cr.rigid_group_positions,
                                    cr.torsion_angle, torch.from_numpy(T["atom14_to_group"])) * tmask[..., None]
        lig_pos = cr.lig_pos.numpy()
        perms = ligand.automorphisms(np.zeros(cr.n_l, int), np.stack([cr.bond_src.numpy(), cr.bond_dst.numpy()]))
        zero = np.zeros(3, np.float32)
        out = pex.pose_metrics(lig_traj, prot_traj, zero, lig_pos, tgt, tmask, seq, perms=perms)
        torch.cuda.synchronize()
        want = oracle_metrics(lig_traj.cpu(), prot_traj.cpu(), zero, lig_pos, tgt.numpy(), tmask.numpy(), seq, perms, np.ones(cr.n_l, bool))
        for k in ("centroid", "sc_rmsd", "lig_rmsd"):
            assert np.abs(out[k].cpu().numpy() - want[k].numpy()).max() < 5e-5, k
        got_rate, want_rate = out["chi_rate"].cpu().numpy(), want["chi_rate"].numpy()
        assert (np.isnan(got_rate) == np.isnan(want_rate)).all()       # a chi no residue of the pocket has: 0 / 0 in both
        assert np.abs(np.nan_to_num(got_rate) - np.nan_to_num(want_rate)).max() < 0.05


@pytest.mark.gpu
def test_gpu_complex_modeling_frame_and_files(tmp_path):
    """The reference-shaped entry point: frame columns / arr_df like export.py:106-312, files equal to the reference text."""
    z = fixture()
    dev = torch.device("cuda:0")
    seq = z["aatype"][z["pocket_mask"]]
    e = pex.ComplexOutput(name="set:3dbs", ligand_traj=torch.from_numpy(z["lig_traj"]).to(dev), protein_traj=torch.from_numpy(z["prot_traj"]).to(dev),
                          pocket_center_pos=z["center"], ligand_pos=z["lig_pos"], ligand_labels=z["lig_elements"],
                          ligand_edge_index=z["lig_edge_index"], topology=topology(z), atom14_position=z["target_atom14"],
                          atom14_mask=z["target_atom14_mask"], aatype=seq, row={"protein": "3dbs_protein.pdb", "ligand": "x.sdf"},
                          heavy_mask=z["ha_mask"])
    written = []
    frame, arr = pex.complex_modeling([e, e], export_dir=tmp_path, calc_metrics=True, lrmsd_naming=True, complex_name_split=":",
                                      export_fullp=True, export_pkt=True,
                                      ligand_writer=lambda ent, i, pos, path: written.append((i, pos.shape, os.path.basename(path))))
    assert list(frame.columns) == ["protein", "ligand", "centroid", "chi1_15", "sc-rmsd", "l-rmsd", "sample_id", "docked_lig", "protein_pdb"]
    assert len(frame) == 8 and arr["centroid"].shape == (2, 4) and set(arr) == {"centroid", "chi1_15", "sc-rmsd", "l-rmsd"}
    assert np.abs(arr["centroid"][0] - z["ref_centroid"][:, -1]).max() < TOL
    assert np.abs(arr["sc-rmsd"][1] - z["ref_sc_rmsd"][:, -1]).max() < TOL
    assert np.abs(arr["l-rmsd"][0] - z["ref_symm_rmsd"][:, -1]).max() < TOL
    assert frame["sample_id"][3] == "sample_4_" + pex.rmsd_to_str(float(arr["l-rmsd"][0][3])) and len(written) == 8
    for pid in (0, 3):
        d = os.path.join(tmp_path, "3dbs", frame["sample_id"][pid])
        assert open(os.path.join(d, "prot_final.pdb")).read() == bytes(z[f"ref_pdb_full_{pid}"]).decode()
        assert open(os.path.join(d, "pkt_final.pdb")).read() == bytes(z[f"ref_pdb_pkt_{pid}"]).decode()
        assert frame["protein_pdb"][pid] == os.path.join(d, "prot_final.pdb")
    with pytest.raises(NotImplementedError):
        pex.complex_modeling([e], export_dir=tmp_path, export_pkt_traj=True)
    frame2, arr2 = pex.complex_modeling([e])
    assert arr2 is None and list(frame2.columns) == ["protein", "ligand"] and len(frame2) == 4
    frame3, arr3 = pex.complex_modeling([e], calc_metrics=True)          # metrics without export: no l-rmsd (export.py:198 precedes :215)
    assert list(frame3.columns) == ["protein", "ligand", "centroid", "chi1_15", "sc-rmsd"] and set(arr3) == {"centroid", "chi1_15", "sc-rmsd"}


# ------------------------------------------------------------------------------------------------ lig_final.sdf
_MOLBLOCK = """glycine zwitterion
  toolkit           3D

 10  9  0  0  0  0  0  0  0  0999 V2000
    0.0000    0.0000    0.0000 N   0  3  0  0  0  0  0  0  0  0  0  0
    1.4500    0.0000    0.0000 C   0  0  0  0  0  0  0  0  0  0  0  0
    0.3300    0.9400    0.0000 H   0  0  0  0  0  0  0  0  0  0  0  0
    2.0000    1.4000    0.0000 C   0  0  0  0  0  0  0  0  0  0  0  0
    1.3000    2.4000    0.0000 O   0  0  0  0  0  0  0  0  0  0  0  0
    3.2500    1.5000    0.0000 O   0  5  0  0  0  0  0  0  0  0  0  0
   -0.3300   -0.4700    0.8100 H   0  0  0  0  0  0  0  0  0  0  0  0
   -0.3300   -0.4700   -0.8100 H   0  0  0  0  0  0  0  0  0  0  0  0
    1.8100   -0.5200    0.8900 H   0  0  0  0  0  0  0  0  0  0  0  0
    1.8100   -0.5200   -0.8900 H   0  0  0  0  0  0  0  0  0  0  0  0
  1  2  1  0
  1  3  1  0
  2  4  1  0
  4  5  2  0
  4  6  1  0
  1  7  1  0
  1  8  1  0
  2  9  1  0
  2 10  1  0
M  CHG  2   1   1   6  -1
M  END
>  <ID>
gly

$$$$
"""


def test_sdf_template_writes_heavy_atom_poses(tmp_path):
    """lig_final.sdf (evaluation/export.py:236-244): the input mol block without its hydrogens (RemoveHs: what the model
    ligand holds), atoms renumbered, charges remapped, the pose's coordinates in %10.4f columns.  Host code of the C ABI
    (dbfr_sdf_format / dbfr_sdf_write_files) against a plain-Python formatting of the same block."""
    t = ligand.SdfTemplate.from_molblock(_MOLBLOCK)
    assert t.n_atoms == 5                                   # N, C, C, O, O
    rng = np.random.default_rng(0)
    poses = rng.normal(0, 30, (7, 5, 3)).astype(np.float32)
    poses[0, 0] = [-123.45678, 0.00004, 9999.5]
    text = t.format(poses[0])
    L_ = text.split("\n")
    assert L_[0] == "glycine zwitterion" and L_[1] == "  DBFR-HIP          3D" and L_[3].startswith("  5  4") and "V2000" in L_[3]
    want_atoms = [f"{x:10.4f}{y:10.4f}{z:10.4f}" for x, y, z in poses[0].astype(np.float64)]
    tails = [" N   0  3  0  0  0  0  0  0  0  0  0  0", " C   0  0  0  0  0  0  0  0  0  0  0  0", " C   0  0  0  0  0  0  0  0  0  0  0  0",
             " O   0  0  0  0  0  0  0  0  0  0  0  0", " O   0  5  0  0  0  0  0  0  0  0  0  0"]
    assert L_[4:9] == [a + b for a, b in zip(want_atoms, tails)]
    assert L_[9:13] == ["  1  2  1  0", "  2  3  1  0", "  3  4  2  0", "  3  5  1  0"]     # bonds to H dropped, renumbered
    assert L_[13] == "M  CHG  2   1   1   5  -1" and L_[14] == "M  END"
    assert L_[15:19] == [">  <ID>", "gly", "", "$$$$"] and text.endswith("$$$$\n")
    paths = [str(tmp_path / f"sample_{i}" / "lig_final.sdf") for i in range(7)]
    for p in paths:
        os.makedirs(os.path.dirname(p))
    t.write_poses(poses, paths, threads=3)
    for i, p in enumerate(paths):
        assert open(p).read() == t.format(poses[i])
    with pytest.raises(Exception):
        t.write_poses(poses[:, :4], paths)
    keep_h = ligand.SdfTemplate.from_molblock(_MOLBLOCK, remove_hs=False)
    assert keep_h.n_atoms == 10 and "M  CHG  2   1   1   6  -1" in keep_h.trailer


def test_sdf_template_first_record_only_and_stale_atom_lines_dropped():
    """a multi-record SD file must not leak its later molecules into every pose's lig_final.sdf, and property lines that
    carry atom numbers in layouts the renumbering does not rewrite (`A  nnn` aliases + their text line, `V  nnn`, `M  RGP` ...) must
    not survive with stale numbers once hydrogens were dropped (kept verbatim when nothing was renumbered)."""
    lines = _MOLBLOCK.split("\n")
    end = lines.index("M  END")
    extra = ["A    4", "carboxyl O", "V    9 a hydrogen value", "M  RGP  1   2   1"]
    two = "\n".join(lines[:end] + extra + lines[end:]) + "\nsecond molecule\n  x\n\n  1  0  0  0  0  0  0  0  0  0999 V2000\n" \
        "    0.0000    0.0000    0.0000 C   0  0\nM  END\n$$$$\n"
    t = ligand.SdfTemplate.from_molblock(two)
    assert t.n_atoms == 5
    tr = t.trailer.split("\n")
    assert not any(l.startswith(("A  ", "V  ", "M  RGP")) or l == "carboxyl O" for l in tr)
    assert "second molecule" not in t.trailer and t.trailer.count("$$$$") == 1 and t.trailer.endswith("$$$$\n")
    assert "M  CHG  2   1   1   5  -1" in tr and ">  <ID>" in tr           # remapped charges and the data item survive
    keep = ligand.SdfTemplate.from_molblock(two, remove_hs=False)            # nothing renumbered: the lines stay as they are
    assert all(x in keep.trailer.split("\n") for x in extra) and "second molecule" not in keep.trailer


def test_chi_differ_matches_the_reference_fixture():
    """tests/golden/chi_differ.npz: what the reference's own `chi_differ` (metrics/angbin.py:48-103) returned for the 3DBS
    poses of export.npz, its `atom37_to_torsion_angles` being the openfold copy the reference vendors
    (make_golden.py::golden_chi_differ).  Pins the restatement to 1e-4 rad (fp32 noise of the two formulations)."""
    z, c = fixture(), np.load(os.path.join(GOLDEN, "chi_differ.npz"))
    seq = torch.from_numpy(z["aatype"][z["pocket_mask"]])
    center = torch.from_numpy(z["center"])
    d, m = oex.chi_differ(torch.from_numpy(z["prot_traj"]) + center, torch.from_numpy(z["target_atom14"]) + center,
                          torch.from_numpy(z["target_atom14_mask"]), seq, T)
    assert torch.equal(m.reshape(-1, 4).bool(), torch.from_numpy(c["ref_mask"]).bool())
    assert (d - torch.from_numpy(c["ref_delta_chi"])).abs().max() < 1e-4
    assert (oex.chi_success_rate(d, m) - torch.from_numpy(c["ref_chi_rate"])).abs().max() < 1e-6


@pytest.mark.gpu
def test_gpu_delta_chi_matches_the_reference_chi_differ():
    """k_pose_metrics' per-residue |delta chi| and chi_1..4 success rates against the REFERENCE's chi_differ output
    (chi_differ.npz, openfold transforms vendored by the reference): 2e-4 rad."""
    z, c = fixture(), np.load(os.path.join(GOLDEN, "chi_differ.npz"))
    dev = torch.device("cuda:0")
    seq = z["aatype"][z["pocket_mask"]]
    out = pex.pose_metrics(torch.from_numpy(z["lig_traj"]).to(dev), torch.from_numpy(z["prot_traj"]).to(dev), z["center"], z["lig_pos"],
                           z["target_atom14"], z["target_atom14_mask"], seq, with_delta_chi=True)
    assert np.abs(out["delta_chi"].cpu().numpy() - c["ref_delta_chi"]).max() < 2e-4
    assert np.abs(out["chi_rate"].cpu().numpy() - c["ref_chi_rate"]).max() < 1e-6
