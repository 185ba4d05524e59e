"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's per-pose initialisation
(SURVEY.md 8(f) row f1): the real-time transforms that turn one per-complex record
into one randomised starting pose, and the batch collation of those poses.

Follows druglib/datasets/Docking/struct_init.py:
  LigInit        :16-53   uniform torsion kicks (no Kabsch), random rotation about the
                          centroid, N(0, tr_sigma_max) translation -- the centroid is NOT added back
  SCFixer        :61-106  residues whose chi mask disagrees with the residue type get the AF2
                          literature frames / positions / atom14 mask
  SCProtInit     :114-138 chi <- U(-pi, pi) * mask, psi kept; atom14 rebuilt from the templates
and druglib/datasets/Docking/formatting.py:41-51 (Atom14ToAllAtomsRepr), :6-26 (PLData
increments used by the collate).

The reference draws its random numbers from the process-global numpy / scipy / torch
generators; here every function takes the drawn numbers as arguments (the "init tape"),
in the reference's draw order per pose:
  1. np.random.uniform(-pi, pi, n_tor)           (only when the ligand has torsions)
  2. scipy Rotation.random().as_matrix()         (3x3)
  3. torch.normal(0, tr_sigma_max, (1, 3))
  4. np.random.uniform(-pi, pi, (n_res, 4))
Pinned against the reference's own struct_init.py run on the same draws, and ``collate``
against the reference's own druglib.data (Batch.from_data_list -> to_dict), by
tests/golden/make_golden.py (bit-exact) -> tests/golden/pose_init.npz.
"""
import numpy as np
import torch

from . import geometry


def lig_init(lig_pos, lig_edge_index, tor_edge_mask, rot_node_mask, torsion_updates, rotation, tr_update):
    """struct_init.py:24-53.  lig_pos [N,3] f32; lig_edge_index [2,E]; tor_edge_mask [E];
    rot_node_mask bool [n_tor,N]; torsion_updates float64 [n_tor] or None; rotation [3,3]
    (float64 as scipy returns it); tr_update [1,3] f32."""
    pos = lig_pos
    n_tor = int(torch.as_tensor(tor_edge_mask).sum())
    if n_tor > 0:
        pos = geometry.modify_conformer_torsion_angles(
            pos, lig_edge_index.T[torch.as_tensor(tor_edge_mask).bool()], rot_node_mask, torsion_updates)
    center = torch.mean(pos, dim=0, keepdim=True)
    R = torch.from_numpy(np.asarray(rotation)).float()
    return (pos - center) @ R.T + tr_update


def sc_fixer(rec, tables):
    """struct_init.py:72-106 on one per-complex record (dict of tensors, modified copy returned).
    tables: chi_mask [21,4], default_frame [21,8,4,4], atom14_lit_pos [21,14,3], atom14_mask [21,14]."""
    rec = dict(rec)
    seq = rec["sequence"]
    sc_mask = rec["sc_torsion_edge_mask"].clone()
    exists = torch.from_numpy(np.asarray(tables["chi_mask"])[seq.numpy()]).bool()
    agree = (exists == sc_mask).all(dim=-1)
    bad = torch.logical_not(agree)
    if int(bad.sum()) == 0:
        return rec
    atom14_mask = rec["atom14_mask"].clone()
    bb_exists = atom14_mask[:, :3].bool().all()          # NB a scalar over ALL residues (reference quirk, :83)
    bad = torch.logical_and(bad, bb_exists)
    sc_mask[bad] = exists[bad].to(sc_mask.dtype)
    rec["sc_torsion_edge_mask"] = sc_mask * bb_exists.unsqueeze(-1)
    res = seq[bad]
    df, rg = rec["default_frame"].clone(), rec["rigid_group_positions"].clone()
    df[bad] = torch.from_numpy(np.asarray(tables["default_frame"])).float()[res]
    rg[bad] = torch.from_numpy(np.asarray(tables["atom14_lit_pos"])).float()[res]
    atom14_mask[bad] = torch.from_numpy(np.asarray(tables["atom14_mask"])).bool()[res].to(atom14_mask.dtype)
    rec["default_frame"], rec["rigid_group_positions"], rec["atom14_mask"] = df, rg, atom14_mask
    return rec


def sc_prot_init(rec, torsion_updates, atom14_to_group):
    """struct_init.py:117-138.  torsion_updates float64 [n_res,4].  Returns (torsion_angle [n_res,5],
    atom14_position [n_res,14,3] already multiplied by atom14_mask)."""
    sc_mask = rec["sc_torsion_edge_mask"]
    ta = rec["torsion_angle"].clone()
    assert ta.shape == (rec["sequence"].shape[0], 5)
    ta[:, 1:] = torch.from_numpy(np.asarray(torsion_updates) * sc_mask.numpy())
    pos14 = geometry.build_atom14(rec["sequence"], rec["backbone_transl"], rec["backbone_rots"], rec["default_frame"],
                                  rec["rigid_group_positions"], ta, atom14_to_group)
    pos14 = pos14 * rec["atom14_mask"].unsqueeze(-1)
    return ta, pos14


def to_all_atoms(rec, atom14_position):
    """formatting.py:46-51: compact atom14 -> heavy-atom list (row-major (res, slot))."""
    m = rec["atom14_mask"].bool()
    return atom14_position[m], rec["pocket_node_feature"][m]


def init_pose(rec, tape, tables):
    """LigInit -> SCProtInit -> Atom14ToAllAtomsRepr on a record that already went through sc_fixer.
    tape: dict(tor [n_tor] f64 | None, rot [3,3], tr [1,3] f32, sc [n_res,4] f64).
    Returns the per-pose dict the collate consumes."""
    out = dict(rec)
    out["lig_pos"] = lig_init(rec["lig_pos"], rec["lig_edge_index"], rec["tor_edge_mask"], rec["rot_node_mask"],
                              tape.get("tor"), tape["rot"], tape["tr"])
    ta, pos14 = sc_prot_init(rec, tape["sc"], tables["atom14_to_group"])
    out["torsion_angle"] = ta
    out["rec_atm_pos"], out["pocket_node_feature"] = to_all_atoms(rec, pos14)
    return out


def collate(poses):
    """druglib/data/collate.py:18-137 with PLData's increments (formatting.py:6-26) and
    follow_batch = ['lig_node', 'rec_atm_pos'] (diffbindfr_ts.py:92-96): concatenate along dim 0
    (lig_edge_index along dim 1, offset by the ligand-atom count; torsion_edge_index along dim 0,
    offset by the pocket heavy-atom count)."""
    out = {}
    lo = ao = 0
    ei, tei, lb, ab = [], [], [], []
    for g, p in enumerate(poses):
        ei.append(p["lig_edge_index"] + lo)
        tei.append(p["torsion_edge_index"] + ao)
        lb.append(torch.full((p["lig_node"].shape[0],), g, dtype=torch.int64))
        ab.append(torch.full((p["rec_atm_pos"].shape[0],), g, dtype=torch.int64))
        lo += p["lig_node"].shape[0]
        ao += p["rec_atm_pos"].shape[0]
    for k in ("lig_node", "lig_pos", "lig_edge_feat", "tor_edge_mask", "pocket_node_feature", "rec_atm_pos",
              "sc_torsion_edge_mask", "backbone_transl", "sequence", "atom14_mask", "backbone_rots", "default_frame",
              "rigid_group_positions", "torsion_angle"):
        out[k] = torch.cat([p[k] for p in poses], 0)
    out["lig_edge_index"] = torch.cat(ei, 1)
    out["torsion_edge_index"] = torch.cat(tei, 0)
    out["lig_node_batch"] = torch.cat(lb)
    out["rec_atm_pos_batch"] = torch.cat(ab)
    ptr = lambda xs: torch.tensor([0] + [x.shape[0] for x in xs], dtype=torch.int64).cumsum(0)
    out["lig_node_ptr"], out["rec_atm_pos_ptr"] = ptr(lb), ptr(ab)
    out["rot_node_mask"] = [p["rot_node_mask"] for p in poses]
    out["num_graphs"] = len(poses)
    return out
