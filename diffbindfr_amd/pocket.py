"""Once-per-pocket preparation: residue types + atom14 coordinates in, the pocket half of a
per-complex record out (the keys of DiffBindFR/configs/diffbindfr_ts.py:49-55 that describe the pocket).

Replaces, for one pocket,
  SCPocketFinderDefault.process_pocket   druglib/datasets/Docking/pocket_pipeline.py:172-189
      -> extract_chi_and_template        druglib/utils/obj/prot_math.py:116-241      (HIP: dbfr_extract_templates)
  PocketGraphBuilder                     pocket_pipeline.py:192-207 -> build_torsion_edges / make_torsion_mask
                                         (datasets/Docking/utils.py:94-126, prot_math.py:350-391)
  PocketFeaturizer                       pocket_pipeline.py:213-262 (the five categorical columns)
  Decentration                           pocket_pipeline.py:276-300 (pocket moved to its CA centroid)
The integer parts are table look-ups done with torch on the device; the numeric part is the HIP kernel.  Many pockets
(config 4: 2 000) go through one launch: pass them concatenated, with ``res_ptr`` marking the pocket boundaries.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .tables import residue_tables


def _tables(dev):
    T = residue_tables()
    t = lambda k, dt: torch.as_tensor(np.asarray(T[k])).to(device=dev, dtype=dt)
    return dict(chi_atoms14=t("chi_atoms14", torch.long), chi_mask=t("chi_mask", torch.float32),
                torsion_edges=t("torsion_edges", torch.long), atom14_mask=t("atom14_mask", torch.bool),
                atom14_to_atom37=t("atom14_to_atom37", torch.long), atom37_to_coarse=t("atom37_to_coarse", torch.float32),
                atom37_to_element=t("atom37_to_element", torch.float32))


@torch.no_grad()
def extract_templates(aatype, atom14_position):
    """``dbfr_extract_templates`` on device tensors: aatype [N] int, atom14_position [N,14,3] (slots the residue type
    does not use are ignored).  Returns dict(backbone_transl, backbone_rots, default_frame, rigid_group_positions,
    torsion_angle [N,5] radians)."""
    lib = L.load()
    dev = atom14_position.device
    if dev.type != "cuda":
        raise L.DbfrError("extract_templates needs a ROCm device (no CPU path)")
    n = int(aatype.shape[0])
    aa = aatype.to(device=dev, dtype=torch.int32).contiguous()
    pos = atom14_position.to(torch.float32).contiguous()
    out = dict(backbone_transl=torch.empty(n, 3, device=dev), backbone_rots=torch.empty(n, 3, 3, device=dev),
               default_frame=torch.empty(n, 8, 4, 4, device=dev), rigid_group_positions=torch.empty(n, 14, 3, device=dev),
               torsion_angle=torch.empty(n, 5, device=dev))
    p = lambda x: C.c_void_p(x.data_ptr())
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.dbfr_extract_templates(n, p(aa), p(pos), p(out["backbone_transl"]), p(out["backbone_rots"]),
                                           p(out["default_frame"]), p(out["rigid_group_positions"]), p(out["torsion_angle"]), stream))
    return out


def torsion_mask(aatype, atom14_mask, tb):
    """chi_k exists for the residue type and all four of its atoms are present (prot_math.py:350-391)."""
    idx = tb["chi_atoms14"][aatype]                                              # [N,4,4]
    atoms = torch.gather(atom14_mask.float()[:, None, :].expand(-1, 4, -1), 2, idx)
    return (tb["chi_mask"][aatype] * atoms.prod(dim=-1)).bool()


def torsion_edges(aatype, atom14_mask, res_ptr, tb):
    """The j-k bond of every chi in per-pocket compacted atom indices (datasets/Docking/utils.py:94-126)."""
    m = atom14_mask.bool()
    flat = m.reshape(-1).long()
    idx = torch.cumsum(flat, 0) - flat                                          # exclusive scan = compacted index
    n_res = m.shape[0]
    # per-pocket base: atoms before the pocket's first residue
    atoms_before_res = torch.cat([idx.new_zeros(1), torch.cumsum(m.sum(1), 0)])[:-1]      # [N]
    pocket_of_res = torch.bucketize(torch.arange(n_res, device=m.device), res_ptr[1:], right=True)
    base = atoms_before_res[res_ptr[:-1]][pocket_of_res]                         # [N]
    node_idx = (idx.reshape(n_res, 14) - base[:, None]) * m
    te = torch.gather(node_idx[:, None, :].expand(-1, 4, -1), 2, tb["torsion_edges"][aatype])
    chis = torsion_mask(aatype, m, tb)
    return te * chis[..., None], chis


def node_features(aatype, atom14_mask, tb):
    a37 = tb["atom14_to_atom37"][aatype]
    bb = torch.zeros(aatype.shape[0], 14, device=aatype.device)
    bb[:, :4] = 1.0
    f = torch.stack([a37.float(), tb["atom37_to_coarse"][a37], tb["atom37_to_element"][a37],
                     aatype[:, None].expand(-1, 14).float(), bb], dim=-1)
    return f * atom14_mask.bool().unsqueeze(-1)


@torch.no_grad()
def pocket_records(aatype, atom14_position, atom14_mask, res_ptr=None, device="cuda:0", decentre=True):
    """Pocket halves of per-complex records for one or many pockets.

    aatype [N]; atom14_position [N,14,3]; atom14_mask [N,14] = atoms actually present (missing side chains allowed:
    SCFixer in ``assemble.ComplexRecord`` completes them); res_ptr [P+1] pocket boundaries (None = one pocket).
    Returns a list of P dicts with the reference's keys (CPU tensors), each ready to be merged with the ligand half
    and handed to ``assemble.ComplexRecord``; ``pocket_center_pos`` is the CA centroid that was subtracted."""
    dev = torch.device(device)
    aa = torch.as_tensor(aatype).to(dev).long()
    pos = torch.as_tensor(atom14_position).to(device=dev, dtype=torch.float32)
    mask = torch.as_tensor(atom14_mask).to(dev).bool()
    n = aa.shape[0]
    rp = torch.tensor([0, n], device=dev) if res_ptr is None else torch.as_tensor(res_ptr).to(dev).long()
    tb = _tables(dev)
    # the templates come from the positions of the type's full atom set (Protein.to_pos14 without the missing-atom
    # mask, prot_math.py:540-550); absent atoms sit at the origin there, like in the reference
    ideal = tb["atom14_mask"][aa]
    pocket_of_res = torch.bucketize(torch.arange(n, device=dev), rp[1:], right=True)
    centre = torch.zeros(rp.numel() - 1, 3, device=dev)
    if decentre:                      # Decentration: masked mean of the CA atoms of each pocket
        w = mask[:, 1].float()
        centre.index_add_(0, pocket_of_res, pos[:, 1] * w[:, None])
        cnt = torch.zeros(rp.numel() - 1, device=dev).index_add_(0, pocket_of_res, w)
        centre = centre / cnt.clamp(min=1.0)[:, None]
    tpl = extract_templates(aa, pos * ideal[..., None])
    te, chis = torsion_edges(aa, mask, rp, tb)
    feat = node_features(aa, mask, tb)
    shift = centre[pocket_of_res]
    out = []
    cpu = lambda x: x.cpu()
    rpl = rp.tolist()
    full = dict(sequence=aa, backbone_transl=tpl["backbone_transl"] - shift, backbone_rots=tpl["backbone_rots"],
                default_frame=tpl["default_frame"], rigid_group_positions=tpl["rigid_group_positions"],
                torsion_angle=tpl["torsion_angle"], torsion_edge_index=te, sc_torsion_edge_mask=chis,
                pocket_node_feature=feat, atom14_mask=mask, atom14_position=(pos - shift[:, None]) * mask[..., None])
    full = {k: cpu(v) for k, v in full.items()}
    centre = centre.cpu()
    for p in range(len(rpl) - 1):
        d = {k: v[rpl[p]:rpl[p + 1]].clone() for k, v in full.items()}
        d["pocket_center_pos"] = centre[p]
        out.append(d)
    return out


def select_pocket(atom_pos, atom_mask, ref_pos, cutoff=12.0, max_neighbors=None, res_ptr=None, ref_ptr=None, atoms_id=None):
    """Binding-site residues on the device -- ``Protein.query_region`` / ``select_bs_any`` / ``select_bs_atoms`` /
    ``select_bs_centroid`` (druglib/utils/obj/protein.py:154-240, druglib/utils/bio_utils/select_pocket.py:12-199; the shipped
    pipeline: mode 'any', cutoff 12 A, all ligand atoms, diffbindfr_ts.py:33-41).

    atom_pos [N_res, M, 3] (atom37 / atom14, or [N_res, 3] centroids), atom_mask [N_res, M] (or [N_res]), ref_pos [N_ref, 3];
    several proteins at once: rows res_ptr[p]..res_ptr[p+1] belong to protein p and use the reference points
    ref_ptr[p]..ref_ptr[p+1].  ``atoms_id``: atom columns to use (the reference's atom-name / element modes).
    Returns (mask bool [N_res], min_dist2 float [N_res]) on the device."""
    lib = L.load()
    dev = atom_pos.device
    if dev.type != "cuda":
        raise L.DbfrError("select_pocket needs a ROCm device (no CPU path)")
    pos = atom_pos.to(torch.float32)
    msk = torch.as_tensor(atom_mask).to(device=dev, dtype=torch.float32)
    if pos.dim() == 2:                                  # centroid mode: one point per residue, present if any atom is
        pos = pos[:, None, :]
        msk = (msk.reshape(pos.shape[0], -1) != 0).any(dim=-1, keepdim=True).float()
    if atoms_id is not None:
        idx = torch.as_tensor(list(atoms_id), device=dev, dtype=torch.long)
        pos, msk = pos[:, idx], msk[:, idx]
    pos, msk = pos.contiguous(), msk.contiguous()
    n, m = int(pos.shape[0]), int(pos.shape[1])
    ref = torch.as_tensor(ref_pos).to(device=dev, dtype=torch.float32).contiguous()
    i32 = lambda x, default: torch.as_tensor(default if x is None else x).to(device=dev, dtype=torch.int32).contiguous()
    rp, fp = i32(res_ptr, [0, n]), i32(ref_ptr, [0, int(ref.shape[0])])
    if rp.numel() != fp.numel():
        raise L.DbfrError("res_ptr and ref_ptr must describe the same number of proteins")
    d2 = torch.empty(n, device=dev)
    out = torch.empty(n, device=dev, dtype=torch.uint8)
    p = lambda x: C.c_void_p(x.data_ptr())
    with torch.cuda.device(dev):
        L.check(lib.dbfr_select_pocket(int(rp.numel()) - 1, n, p(rp), m, p(pos), p(msk), p(fp), p(ref), float(cutoff),
                                       0 if max_neighbors is None else int(max_neighbors), p(d2), p(out),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out.bool(), d2


@torch.no_grad()
def atom37_to_atom14(aatype, atom37_pos, atom37_mask):
    """``Protein.to_pos14(consider_missing_atoms=True)`` (druglib/utils/obj/prot_math.py:18-43) on the device: the atom14 slots of
    every residue gathered from its atom37 row, masked by (atom present) x (slot used by the residue type).
    Returns (atom14_pos [N,14,3], atom14_mask bool [N,14])."""
    dev = atom37_pos.device
    tb = _tables(dev)
    aa = torch.as_tensor(aatype).to(dev).long()
    m = tb["atom14_to_atom37"][aa]                                                  # [N,14] atom37 id of each slot
    mask14 = torch.gather(torch.as_tensor(atom37_mask).to(dev).float(), 1, m) * tb["atom14_mask"][aa].float()
    pos14 = torch.gather(atom37_pos.float(), 1, m[..., None].expand(-1, -1, 3)) * mask14[..., None]
    return pos14, mask14.bool()


@torch.no_grad()
def pockets_from_proteins(aatype, atom37_pos, atom37_mask, ref_pos, cutoff=12.0, max_neighbors=None, res_ptr=None, ref_ptr=None,
                          decentre=True):
    """The protein side of ``test_pre_transform_prot`` after ``LoadProtein`` (diffbindfr_ts.py:33-45): SCPocketFinderDefault
    (selection by the ligand's atoms + template extraction), PocketGraphBuilder, PocketFeaturizer, Decentration -- for any
    number of (protein, reference ligand) pairs in one pass.  Returns (records: list of pocket halves as ``pocket_records``
    gives them, pocket_mask bool [N_res_total] on the device)."""
    dev = atom37_pos.device
    mask, _ = select_pocket(atom37_pos, atom37_mask, ref_pos, cutoff, max_neighbors, res_ptr, ref_ptr)
    n = int(mask.shape[0])
    rp = torch.tensor([0, n], device=dev) if res_ptr is None else torch.as_tensor(res_ptr).to(dev).long()
    csum = torch.cat([mask.new_zeros(1, dtype=torch.long), torch.cumsum(mask.long(), 0)])
    pocket_ptr = csum[rp]                                                           # residues kept before each protein
    keep = torch.nonzero(mask).flatten()
    aa = torch.as_tensor(aatype).to(dev).long()[keep]
    pos14, m14 = atom37_to_atom14(aa, atom37_pos[keep], torch.as_tensor(atom37_mask).to(dev)[keep])
    return pocket_records(aa, pos14, m14, pocket_ptr, device=dev, decentre=decentre), mask
