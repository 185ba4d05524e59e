"""CPU restatement of the two geometry updates of the reverse-diffusion step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned against the reference's
own source (which imports in the build container with shims) by
tests/golden/make_golden.py -> tests/golden/geometry.npz.

Follows:
* druglib/utils/geometry_utils/utils.py:1056-1092 (axis_angle_to_quaternion),
  :672-720 (normalised_quaternion, quaternion_to_rot), :1229-1239 (axis_angle_to_rot)
* druglib/utils/geometry_utils/superimposition.py:375-410 (Kabsch)
* druglib/utils/bio_utils/conformer_utils.py:305-355 (torsion + rigid update),
  :420-473 (update_batchlig_pos; per-ligand python loop kept as in the reference)
* druglib/utils/obj/prot_math.py:243-291 -> geometry_utils/aaframe.py:730-994
  (build_pdb_from_template: AF2 Alg. 24 with per-residue template frames)
* druglib/utils/torch_utils/msc.py:295-310 (robust_normalize)
"""
import torch


def axis_angle_to_quaternion(axis_angle):
    ang = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = ang * 0.5
    s = torch.empty_like(ang)
    small = ang.abs() < 1e-6
    s[~small] = torch.sin(half[~small]) / ang[~small]
    s[small] = 0.5 - (ang[small] * ang[small]) / 48
    return torch.cat([torch.cos(half), axis_angle * s], dim=-1)


def quaternion_to_rot(q):
    sq = (q ** 2).sum(-1, keepdim=True)
    q = q / torch.sqrt(sq)
    w, x, y, z = q.unbind(-1)
    rows = [
        w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y,
        2 * x * y + 2 * w * z, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
        2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, w * w - x * x - y * y + z * z,
    ]
    return torch.stack(rows, dim=-1).reshape(q.shape[:-1] + (3, 3))


def axis_angle_to_rot(axis_angle):
    return quaternion_to_rot(axis_angle_to_quaternion(axis_angle))


def kabsch(A, B):
    """A, B: [3, N]; returns R [3,3], t [3,1] with R A + t ~ B."""
    ca, cb = A.mean(dim=1, keepdim=True), B.mean(dim=1, keepdim=True)
    H = (A - ca) @ (B - cb).T
    U, S, Vt = torch.linalg.svd(H)
    R = Vt.T @ U.T
    if torch.linalg.det(R) < 0:
        R = (Vt.T @ torch.diag(torch.tensor([1.0, 1.0, -1.0]))) @ U.T
    t = -R @ ca + cb
    return R, t


def modify_conformer_torsion_angles(pos, edge_index, rot_node_mask, torsion_updates):
    """edge_index [n_tor, 2] local (u, v); rot_node_mask bool [n_tor, N]."""
    pos = pos.clone()
    for i, e in enumerate(edge_index):
        if torsion_updates[i] == 0:
            continue
        u, v = int(e[0]), int(e[1])
        assert not rot_node_mask[i, u] and rot_node_mask[i, v]
        rot_vec = pos[u] - pos[v]
        rot_vec = rot_vec * torsion_updates[i] / torch.linalg.norm(rot_vec)
        rot_mat = axis_angle_to_rot(rot_vec)
        m = rot_node_mask[i]
        pos[m] = (pos[m] - pos[v]) @ rot_mat.T + pos[v]
    return pos


def modify_conformer(pos, edge_index, tor_edge_mask, rot_node_mask, tr_update, rot_update, torsion_updates=None):
    center = pos.mean(dim=0, keepdim=True)
    rot_mat = axis_angle_to_rot(rot_update.squeeze())
    rigid = (pos - center) @ rot_mat.T + tr_update + center
    if torsion_updates is None:
        return rigid
    flex = modify_conformer_torsion_angles(rigid, edge_index.T[tor_edge_mask], rot_node_mask, torsion_updates)
    R, t = kabsch(flex.T, rigid.T)
    return flex @ R.T + t.T


def update_batchlig_pos(tr_update, rot_update, torsion_updates, pos, edge_index, tor_edge_mask,
                        rot_node_mask, batch):
    """conformer_utils.py:420-473: python loop over ligands (kept: this loop is part
    of what the reference's CPU path costs)."""
    G = int(batch.max().item()) + 1
    ptr = torch.zeros(G + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(torch.bincount(batch, minlength=G), 0)
    tmask = tor_edge_mask.bool()
    edge_batch = batch[edge_index[0]]
    n_tor = torch.bincount(edge_batch[tmask], minlength=G)
    tor_ptr = torch.cat([n_tor.new_zeros(1), n_tor.cumsum(0)])
    out = []
    for g in range(G):
        em = edge_batch == g
        ei = edge_index[:, em] - ptr[g]
        upd = None
        if n_tor[g] > 0:
            upd = torsion_updates[tor_ptr[g]:tor_ptr[g + 1]]
        out.append(modify_conformer(pos[ptr[g]:ptr[g + 1]], ei, tmask[em], rot_node_mask[g].bool(),
                                    tr_update[g:g + 1], rot_update[g], upd))
    return torch.cat(out, dim=0)


# --------------------------------------------------------------------------- side chains
def _compose(R1, t1, R2, t2):
    return R1 @ R2, t1 + (R1 @ t2.unsqueeze(-1)).squeeze(-1)


def build_atom14(sequence, backbone_transl, backbone_rots, default_frame, rigid_group_positions,
                 torsion_angle, atom14_to_group):
    """torsion_angle [N,5] radians (psi, chi1..4) -> atom14 positions [N,14,3].

    ``atom14_to_group`` is restype_atom14_to_rigid_group [21,14] (protein_constants.py:1177).
    Mirrors scFlex.py:212-224 -> prot_math.py:243-291 -> aaframe.py:821-994: (sin,cos)
    -> robust_normalize(eps=1e-6) -> Rx; prepend zero (omega, phi) and identity bb;
    all = default_frame * Rx; chain chi2..4; compose with the backbone frame; place atoms.
    """
    N = sequence.shape[0]
    sc = torch.stack([torch.sin(torsion_angle), torch.cos(torsion_angle)], dim=-1)      # [N,5,2]
    sc = torch.cat([torch.zeros(N, 2, 2), sc], dim=1)                                   # omega, phi = 0
    ident = torch.tensor([[0.0, 1.0]]).expand(N, 1, 2)
    sc = torch.cat([ident, sc], dim=1)                                                  # [N,8,2]
    sc = sc / sc.norm(p=2, dim=-1, keepdim=True).clamp(1e-6)
    sin, cos = sc[..., 0], sc[..., 1]
    Rx = torch.zeros(N, 8, 3, 3)
    Rx[..., 0, 0] = 1.0
    Rx[..., 1, 1] = cos
    Rx[..., 1, 2] = -sin
    Rx[..., 2, 1] = sin
    Rx[..., 2, 2] = cos
    Rd, td = default_frame[..., :3, :3], default_frame[..., :3, 3]
    R_all = Rd @ Rx                                  # rot_x carries zero translation
    t_all = td.clone()
    Rc, tc = [R_all[:, k] for k in range(8)], [t_all[:, k] for k in range(8)]
    for k in (5, 6, 7):
        Rc[k], tc[k] = _compose(Rc[k - 1], tc[k - 1], R_all[:, k], t_all[:, k])
    R8, t8 = torch.stack(Rc, 1), torch.stack(tc, 1)
    Rg = backbone_rots[:, None] @ R8
    tg = backbone_transl[:, None] + (backbone_rots[:, None] @ t8.unsqueeze(-1)).squeeze(-1)
    grp = atom14_to_group[sequence]                  # [N,14]
    idx = grp[..., None, None].expand(N, 14, 3, 3)
    Ra = torch.gather(Rg, 1, idx)
    ta = torch.gather(tg, 1, grp[..., None].expand(N, 14, 3))
    return (Ra @ rigid_group_positions.unsqueeze(-1)).squeeze(-1) + ta
