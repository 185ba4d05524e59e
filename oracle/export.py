"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the output side of the sampler:
the per-pose metrics `complex_modeling` computes over the trajectories and the PDB text it writes per pose.

Follows
  DiffBindFR/common/inference_dataset.py:57-63   add_center_pos
  DiffBindFR/metrics/centroid.py:6-14            calc_lig_centroid
  DiffBindFR/metrics/scrmsd.py:8-89              make_altern_atom14, sidechain_rmsd
  DiffBindFR/metrics/angbin.py:11-103            angular_difference, chi_differ
  DiffBindFR/metrics/lrmsd.py:287-335            symm_rmsd (the automorphism list is an input here)
  DiffBindFR/evaluation/export.py:106-312        complex_modeling: chi1_15 success rate (:176-181), prot_final / pkt_final
  druglib/utils/obj/prot_math.py:18-43,294-316   to_pos14, atom14_to_atom37
  druglib/utils/obj/protein.py:478-537,658-800   Protein.pos_update (atom14 -> atom37), int_id_to_str_id, to_pdb
Pinned against the reference's own functions by tests/golden/make_golden.py -> tests/golden/export.npz, EXCEPT the
dihedral extraction inside chi_differ: the reference calls atom37_to_torsion_angles of the openfold copy it vendors
(/root/reference/openfold/data/data_transforms.py:943-); restated below (chi part only, on the atom14 layout) and PINNED since
round 2: tests/golden/make_golden.py::golden_chi_differ runs the reference's own chi_differ on the 3DBS poses (mask equal,
|delta chi| within 4.8e-5 rad = the fp32 noise of the two formulations, success rates equal) -> tests/golden/chi_differ.npz.
"""
import numpy as np
import torch


def add_center_pos(pos, center):
    return pos + center.view(*((1,) * (pos.dim() - 1) + (3,)))


def calc_lig_centroid(pred_pos, target_pos):
    """pred [..., N, 3], target [N, 3] -> [...] distance between centroids."""
    return (pred_pos.mean(dim=-2) - target_pos.mean(dim=-2)).norm(dim=-1)


def make_altern_atom14(atom14_pos, atom14_mask, sequence, tables):
    """scrmsd.py:8-61: the renaming matrices are permutations, so the einsum is a gather by the swap table."""
    swap = torch.as_tensor(np.asarray(tables["atom14_swap"])).long()[sequence]          # [N,14]
    idx = swap.view(*((1,) * (atom14_pos.dim() - 3) + swap.shape))
    alt_pos = torch.gather(atom14_pos, -2, idx[..., None].expand(*atom14_pos.shape[:-3], *swap.shape, 3))
    alt_mask = torch.gather(atom14_mask.float(), -1, idx.expand(*atom14_mask.shape[:-2], *swap.shape))
    return alt_pos, alt_mask


def sidechain_rmsd(pred_atom14, target_atom14, target_atom14_mask, sequence, tables, eps=1e-6):
    """scrmsd.py:64-89.  pred [..., N, 14, 3] -> [...] mean over residues with a side chain of the per-residue RMSD,
    minimum over the two namings of the 180-degree-symmetric groups."""
    lead = (1,) * (pred_atom14.dim() - 3)
    tgt = target_atom14.view(*lead, *target_atom14.shape)
    msk = target_atom14_mask.view(*lead, *target_atom14_mask.shape)
    sc_mask = msk[..., 5:]
    sc_pred = pred_atom14[..., 5:, :] * sc_mask[..., None]
    sc_tgt = tgt[..., 5:, :] * sc_mask[..., None]
    alt_pos, alt_mask = make_altern_atom14(tgt, msk, sequence, tables)
    alt_tgt = alt_pos[..., 5:, :] * alt_mask[..., 5:][..., None]
    d2 = ((sc_tgt - sc_pred) ** 2).sum(dim=(-2, -1))
    d2_alt = ((alt_tgt - sc_pred) ** 2).sum(dim=(-2, -1))
    d2 = torch.minimum(d2, d2_alt)
    res_mask = sc_mask.any(dim=-1)
    deno = sc_mask.sum(dim=-1)
    rmsd = torch.sqrt(d2 / (deno + eps)) * res_mask
    return rmsd.sum(dim=-1) / res_mask.sum(dim=-1)


def chi_sin_cos(atom14_pos, atom14_mask, sequence, tables):
    """The chi part of atom37_to_torsion_angles (openfold/data/data_transforms.py:943-1090, vendored by the reference), evaluated on the
    atom14 layout: for every chi_k with dihedral atoms (a0,a1,a2,a3) build the frame with origin a2, x along a2-a1 and
    a0 in the xy plane (eps 1e-8 under both square roots), express a3 in it and return (sin, cos) = (z, y) / sqrt(z^2 +
    y^2 + 1e-8).  Returns (sin_cos [..., N, 4, 2], alt_sin_cos, mask [..., N, 4]); alt flips the pi-periodic chis."""
    idx = torch.as_tensor(np.asarray(tables["chi_atoms14"])).long()[sequence]               # [N,4,4]
    chis = torch.as_tensor(np.asarray(tables["chi_mask"], dtype=np.float32))[sequence]        # [N,4]
    period = torch.as_tensor(np.asarray(tables["chi_pi_periodic"], dtype=np.float32))[sequence]
    lead = atom14_pos.shape[:-3]
    n = idx.shape[0]
    flat = idx.reshape(n, 16)
    g = flat.view(*((1,) * len(lead)), n, 16, 1).expand(*lead, n, 16, 3)
    p = torch.gather(atom14_pos, -2, g).reshape(*lead, n, 4, 4, 3)
    gm = flat.view(*((1,) * (atom14_mask.dim() - 2)), n, 16).expand(*atom14_mask.shape[:-2], n, 16)
    am = torch.gather(atom14_mask.float(), -1, gm).reshape(*atom14_mask.shape[:-2], n, 4, 4)
    mask = chis * torch.prod(am, dim=-1)
    a0, a1, a2, a3 = p[..., 0, :], p[..., 1, :], p[..., 2, :], p[..., 3, :]
    e0 = a2 - a1
    e1 = a0 - a2
    e0 = e0 / torch.sqrt((e0 * e0).sum(-1, keepdim=True) + 1e-8)
    e1 = e1 - e0 * (e0 * e1).sum(-1, keepdim=True)
    e1 = e1 / torch.sqrt((e1 * e1).sum(-1, keepdim=True) + 1e-8)
    e2 = torch.cross(e0, e1, dim=-1)
    rel = a3 - a2
    sc = torch.stack([(e2 * rel).sum(-1), (e1 * rel).sum(-1)], dim=-1)
    sc = sc / torch.sqrt((sc * sc).sum(-1, keepdim=True) + 1e-8)
    alt = sc * (1.0 - 2.0 * period)[..., None]
    return sc, alt, mask


def angular_difference(target, predicted):
    """angbin.py:11-29 as written: fmod keeps the dividend's sign, so a wrapped difference below -pi comes out as pi."""
    t = torch.atan2(target[..., 0], target[..., 1])
    p = torch.atan2(predicted[..., 0], predicted[..., 1])
    d = torch.fmod(p - t + torch.pi, 2. * torch.pi) - torch.pi
    return torch.clamp(torch.abs(d), min=0, max=torch.pi)


def chi_differ(pred_atom14, target_atom14, target_atom14_mask, sequence, tables):
    """angbin.py:48-103.  NB the argument order at :98-99 is angular_difference(pred, target): 'target_angles' there is
    the prediction.  -> (delta [..., N, 4] radians, mask [N, 4] broadcast to the leading dims of pred)."""
    t_sc, t_alt, mask = chi_sin_cos(target_atom14, target_atom14_mask, sequence, tables)
    p_sc, _, _ = chi_sin_cos(pred_atom14, target_atom14_mask, sequence, tables)
    lead = (1,) * (pred_atom14.dim() - 3)
    t_sc, t_alt, mask = t_sc.view(*lead, *t_sc.shape), t_alt.view(*lead, *t_alt.shape), mask.view(*lead, *mask.shape)
    d = torch.minimum(angular_difference(p_sc, t_sc), angular_difference(p_sc, t_alt))
    return d * mask, mask


def chi_success_rate(delta_chi, mask, upper=15 / 180 * torch.pi):
    """export.py:176-179: fraction of existing chi_k below the bound, per leading index -> [..., 4]."""
    crit = (delta_chi < upper) * mask.bool()
    return crit.sum(dim=-2) / mask.sum(dim=-2)


def symm_rmsd(perms, ha_mask, target_pos, pred_pos):
    """lrmsd.py:311-335 after the graph matching: perms = list of (ind, tar) index arrays (tar is arange).
    pred [..., N, 3] -> [...] minimum over the automorphisms of the heavy-atom RMSD."""
    out = []
    pred = torch.as_tensor(pred_pos)
    for ind, tar in perms:
        co = ha_mask[ind] & ha_mask
        i, t = torch.as_tensor(ind[co]).long(), tar[co]
        tp = torch.as_tensor(target_pos[t])
        pp = pred.index_select(-2, i)
        diff = pp - tp.view(*((1,) * (pp.dim() - 2)), *tp.shape)
        out.append(torch.sqrt(diff.square().sum(dim=-1).mean(dim=-1)))
    return torch.amin(torch.stack(out, dim=0), dim=0)


# ------------------------------------------------------------------------------------------------ PDB text

def to_pos14(aatype, atom37_pos, atom37_exists, tables):
    """prot_math.py:18-43 with the missing-atom mask (Protein.to_pos14(True), export.py:262)."""
    n = aatype.shape[0]
    m = np.asarray(tables["atom14_to_atom37"])[aatype]
    rows = np.arange(n).reshape(-1, 1)
    mask14 = (atom37_exists[rows, m] * np.asarray(tables["atom14_mask"])[aatype]).reshape(n, 14, 1)
    return atom37_pos[rows, m] * mask14, mask14


def pos14_to_pos37(aatype, pos14, atom37_exists, tables):
    """protein.py:514-518 (Protein.pos_update with an atom14 array)."""
    n = aatype.shape[0]
    m = np.asarray(tables["atom37_to_atom14"])[aatype]
    return pos14[np.arange(n).reshape(-1, 1), m] * atom37_exists.reshape(-1, 37, 1)


def int_id_to_str_id(num):
    """protein.py:658-676: 1 -> A ... 26 -> Z, 27 -> AA, 28 -> BA (reverse spreadsheet naming)."""
    num -= 1
    out = []
    while num >= 0:
        out.append(chr(num % 26 + ord('A')))
        num = num // 26 - 1
    return ''.join(out)


def to_pdb(aatype, atom37_pos, atom37_exists, residue_index, chain_index, b_factors, tables, remark=None, model=None,
           add_end=True):
    """protein.py:678-800.  remark = the first line (the reference writes 'REMARK   1 CREATED WITH MDLDruglib <version>,
    <today>' when the protein carries none).  The disordered-residue bookkeeping at :716-727 tests `resid not in dict`
    against tuple keys, so no insertion code is ever produced; kept that way."""
    names37 = [str(a) for a in tables["atom37_names"]]
    res3 = [str(a) for a in tables["restype_names3"]]
    lines = []
    if remark is not None:
        lines.append(remark)
    n = aatype.shape[0]
    serial = 1
    for i in range(n):
        tag = int_id_to_str_id(int(chain_index[i]) + 1)
        rn = res3[int(aatype[i])]
        ri = int(residue_index[i])
        for a in range(37):
            if atom37_exists[i, a] < 0.5:
                continue
            atn = names37[a]
            name = atn if len(atn) == 4 else ' ' + atn
            x, y, z = (float(v) for v in atom37_pos[i, a])
            lines.append(f"{'ATOM':<6}{serial:>5} {name:<4}{'':>1}{rn:>3} {tag:>1}{ri:>4}{'':>1}   "
                         f"{x:>8.3f}{y:>8.3f}{z:>8.3f}{1.0:>6.2f}{float(b_factors[i, a]):>6.2f}          {atn[0]:>2}{'':>2}")
            serial += 1
        if i == n - 1 or chain_index[i] != chain_index[i + 1]:
            lines.append(f"{'TER':<6}{serial:>5}      {rn:>3} {tag:>1}{ri:>4}")
            serial += 1
    if model is not None:
        lines.append('ENDMDL')
    if add_end:
        lines.append('END')
    return '\n'.join(line.ljust(80) for line in lines) + '\n'


def pose_pdb(aatype, base_pos37, atom37_exists, residue_index, chain_index, b_factors, pocket_rows, pocket_pos14, tables,
             remark=None):
    """export.py:261-268: the full protein with the pocket residues' atom14 coordinates replaced by one pose."""
    fp14, _ = to_pos14(aatype, base_pos37, atom37_exists, tables)
    fp14 = fp14.copy()
    fp14[pocket_rows] = pocket_pos14
    return to_pdb(aatype, pos14_to_pos37(aatype, fp14, atom37_exists, tables), atom37_exists, residue_index, chain_index,
                  b_factors, tables, remark=remark)
