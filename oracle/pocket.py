"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the once-per-pocket preparation
(SURVEY.md 8(f) row f2): from the pocket's residue types and atom14 coordinates to the
template / frame / torsion / feature tensors the sampler consumes.

Follows
  druglib/utils/obj/prot_math.py:116-241   extract_chi_and_template (numpy)
  druglib/utils/geometry_utils/utils.py:32-148,191-205   rot_vec_around_x_axis, parse_xrot_angle,
                                            make_rigid_transformation_4x4, residue_frame, apply_inv_euclidean
  druglib/utils/obj/prot_math.py:350-391   make_torsion_mask
  druglib/datasets/Docking/utils.py:94-126 build_torsion_edges  (+ pocket_pipeline.py:198-207)
  druglib/datasets/Docking/pocket_pipeline.py:213-262 PocketFeaturizer (the five categorical columns)
  druglib/datasets/Docking/pocket_pipeline.py:276-300 Decentration
Pinned against the reference's own functions by tests/golden/make_golden.py -> tests/golden/pocket.npz.
The numeric part keeps the reference's dtype flow (float32 arrays, float64 where numpy promotes).
"""
import numpy as np
import torch


def rot_vec_around_x_axis(x, radian):
    sin_cos = np.stack([np.sin(radian), np.cos(radian)], axis=-1)
    tmpl = np.array([[0., 0., 0., 0., 0., -1, 0., 1., 0.], [0., 0., 0., 0., 1., 0., 0., 0., 1.]], dtype=sin_cos.dtype)
    rot = np.matmul(sin_cos, tmpl).reshape(-1, 3, 3)
    rot[..., 0, 0] = 1
    return np.einsum('ikl,il->ik' if x.ndim == 2 else 'ikl,iml->imk', rot, x)


def parse_xrot_angle(x):
    n = x.shape[0]
    x_axis = np.zeros((n, 3), dtype=np.float32)
    x_axis[:, 0] = 1.
    x_vec = np.sum(x * x_axis, axis=-1, keepdims=True) * x_axis
    yz_vec = x - x_vec
    yz_norm = np.linalg.norm(yz_vec, axis=-1)
    proj = np.zeros((n, 3), dtype=np.float32)
    proj[:, 0] = x_vec[:, 0]
    proj[:, 1] = yz_norm
    return proj, np.arctan2(yz_vec[:, -1], yz_vec[:, -2])


def make_rigid_transformation_4x4(ex, ey, translation, eps=1e-6):
    exn = ex / (np.linalg.norm(ex, axis=-1, keepdims=True) + eps)
    eyn = ey - np.sum(ey * exn, axis=-1, keepdims=True) * exn
    eyn /= np.linalg.norm(eyn, axis=-1, keepdims=True)
    ez = np.cross(exn, eyn)
    m = np.zeros((ex.shape[0], 4, 4), dtype=np.float32)
    m[:, :, :3] = np.stack([exn, eyn, ez, translation], axis=1)
    m[:, 3, 3] = 1.
    return m.transpose(0, 2, 1).astype(np.float32)


def residue_frame(origin, x_axis, xy_plane, eps=1e-20):
    e0 = x_axis - origin
    e1 = xy_plane - origin
    e0 = e0 / np.sqrt(np.sum(e0 ** 2, axis=-1, keepdims=True) + eps)
    e1 = e1 - e0 * np.sum(e0 * e1, axis=-1, keepdims=True)
    e1 = e1 / np.sqrt(np.sum(e1 ** 2, axis=-1, keepdims=True) + eps)
    e2 = np.cross(e0, e1)
    return np.stack([e0, e1, e2], axis=-1), origin


def apply_inv_euclidean(x, R, T):
    return np.einsum('ilk,iml->imk', R, x - T.reshape(-1, 1, 3))


def extract_chi_and_template(aatype, atom_positions, restype_atom14_mask, tables):
    """prot_math.py:116-241 with return_radian=True.  aatype [N]; atom_positions [N,14,3] f32;
    restype_atom14_mask [N,14,1] (the residue type's ideal mask, to_pos14 without missing atoms).
    tables: chi_atoms14 [21,4,4], chi_mask [21,4], atom14_to_group [21,14]."""
    n = aatype.shape[0]
    chi_to_a14 = np.asarray(tables["chi_atoms14"])[aatype]
    chi_mask = np.asarray(tables["chi_mask"])[aatype]
    a14_group = np.asarray(tables["atom14_to_group"])[aatype]
    tmpl = np.zeros([n, 14, 3], dtype=np.float32)
    frame = np.zeros([n, 8, 4, 4], dtype=np.float32)
    angle = np.zeros([n, 5], dtype=np.float32)
    rots, transl = residue_frame(atom_positions[:, 1], atom_positions[:, 2], atom_positions[:, 0])
    bb = apply_inv_euclidean(atom_positions, rots, transl)
    tmpl[:, 0, :2] = bb[:, 0, :2]          # N
    tmpl[:, 2, :1] = bb[:, 2, :1]          # C
    tmpl[:, 4, :] = bb[:, 4, :]            # CB
    frame[:, 0] = np.eye(4)
    frame[:, 1] = np.eye(4)
    frame[:, 2] = make_rigid_transformation_4x4(ex=tmpl[:, 0] - tmpl[:, 1], ey=np.tile(np.array([1.0, 0.0, 0.0]), (n, 1)),
                                               translation=tmpl[:, 0, :])
    frame[:, 3] = make_rigid_transformation_4x4(ex=tmpl[:, 2] - tmpl[:, 1], ey=tmpl[:, 1] - tmpl[:, 0], translation=tmpl[:, 2])
    psi_frame = apply_inv_euclidean(bb, frame[:, 3, :3, :3], tmpl[:, 2])
    o_pos, psi = parse_xrot_angle(psi_frame[:, 3])
    tmpl[:, 3] = o_pos
    angle[:, 0] = psi
    before = bb                              # NB aliases bb, updated in place like the reference (:225)
    for k in range(4):
        mk = chi_mask[:, k]
        if np.sum(mk) == 0:
            continue
        mk = mk.astype('bool')
        sub14 = before[mk]
        ns = sub14.shape[0]
        mapper = chi_to_a14[mk, k]
        sub4 = sub14[np.arange(ns).reshape(-1, 1), mapper]
        if k == 0:
            mat = make_rigid_transformation_4x4(ex=sub4[:, 2] - sub4[:, 1], ey=sub4[:, 0] - sub4[:, 1], translation=sub4[:, 2])
        else:
            mat = make_rigid_transformation_4x4(ex=sub4[:, 2], ey=np.tile(np.array([-1.0, 0.0, 0.0], dtype=np.float32), (ns, 1)),
                                                translation=sub4[:, 2])
        frame[mk, 4 + k] = mat
        local = apply_inv_euclidean(sub14, mat[:, :3, :3], sub4[:, 2])
        nxt4 = local[np.arange(ns).reshape(-1, 1), mapper]
        _, chi = parse_xrot_angle(nxt4[:, 3])
        angle[mk, k + 1] = chi
        local_rot = rot_vec_around_x_axis(local, -chi)
        gmask = np.where(a14_group[mk] == (k + 4))
        add = np.zeros((ns, 14, 3), dtype=np.float32)
        add[gmask] = local_rot[gmask]
        tmpl[mk] = tmpl[mk] + add
        before[mk] = local_rot
    return dict(sequence=aatype, backbone_transl=transl, backbone_rots=rots, default_frame=frame,
                rigid_group_positions=tmpl * restype_atom14_mask, torsion_angle=angle)


def make_torsion_mask(aatype, atom14_mask, tables):
    """prot_math.py:350-391 for the atom14 representation: chi_k exists for the residue type AND its 4 atoms exist."""
    idx = torch.as_tensor(np.asarray(tables["chi_atoms14"])).long()[aatype]              # [N,4,4]
    chis = torch.as_tensor(np.asarray(tables["chi_mask"], dtype=np.float32))[aatype]       # [N,4]
    m = atom14_mask.float()
    atoms = torch.gather(m[:, None, :].expand(-1, 4, -1), 2, idx)
    return (chis * torch.prod(atoms, dim=-1)).bool()


def build_torsion_edges(aatype, atom14_mask, tables):
    """datasets/Docking/utils.py:94-126 + pocket_pipeline.py:203-206: the j-k bond of every chi in compacted atom
    indices, zeroed where the chi does not exist.  Returns (torsion_edge_index [N,4,2], sc_torsion_edge_mask [N,4])."""
    m = atom14_mask.bool()
    node_idx = torch.zeros(m.shape, dtype=torch.long)
    node_idx[m] = torch.arange(int(m.sum()), dtype=torch.long)
    te = torch.as_tensor(np.asarray(tables["torsion_edges"])).long()[aatype]                # [N,4,2] atom14 slots
    te = torch.gather(node_idx[:, None, :].expand(-1, 4, -1), 2, te)
    chis = make_torsion_mask(aatype, m, tables)
    return te * chis[..., None], chis


def pocket_features(aatype, atom14_mask, tables):
    """pocket_pipeline.py:224-262 without the optional secondary-structure column: (atom37 id, coarse22 id, element4 id,
    aatype, is_backbone) per atom14 slot, times the mask.  [N,14,5] float."""
    a37 = torch.as_tensor(np.asarray(tables["atom14_to_atom37"])).long()[aatype]
    cols = [a37.float(), torch.as_tensor(np.asarray(tables["atom37_to_coarse"]))[a37].float(),
            torch.as_tensor(np.asarray(tables["atom37_to_element"]))[a37].float(), aatype[:, None].repeat(1, 14).float()]
    bb = torch.zeros(aatype.shape[0], 14)
    bb[:, :4] = 1.
    return torch.stack(cols + [bb], dim=-1).float() * atom14_mask.bool().unsqueeze(-1)


def select_bs(ref_pos, atom_positions, atom_mask, cutoff=10.0, max_neighbors=None, big_value=1e20):
    """druglib/utils/bio_utils/select_pocket.py:12-99 (`select_bs` with `_select_min` and `_max_neig_trunc`), the rule behind
    `Protein.query_region` (protein.py:154-240) / `SCPocketFinderDefault` (pocket_pipeline.py:147-161; the shipped config uses mode
    'any', cutoff 12, all ligand atoms).  ref_pos [N_l,3], atom_positions [N_res,M,3], atom_mask [N_res,M] ->
    (bool [N_res], per-residue minimum squared distance).  A residue is in iff its nearest present atom is within the cutoff
    (inclusive) of some reference point; the nearest residue is always in; with max_neighbors only that many nearest stay."""
    m = atom_mask.bool()
    d = torch.sum((atom_positions[..., None, :] - ref_pos[None, None, :, :]) ** 2, dim=-1)          # [N_res,M,N_l]
    d = torch.maximum(d, (~m)[..., None].float() * big_value)
    per_res = torch.amin(d, dim=(-2, -1))
    mask = per_res <= cutoff ** 2
    mask[torch.argmin(per_res)] = True
    if max_neighbors is not None:
        dd = per_res.clone()
        dd[~mask] = big_value
        keep = torch.zeros_like(mask)
        keep[torch.sort(dd)[1][:max_neighbors]] = True
        mask = mask & keep
    return mask, per_res
