"""Shared helpers of the parity tests (test infrastructure; may import oracle/)."""
import copy
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden_batch(path=None):
    """The small collated batch frozen in tests/golden/sampler.npz -> namespace of tensors."""
    z = np.load(path or os.path.join(GOLDEN, "sampler.npz"))
    d = SimpleNamespace()
    G = int(z["num_graphs"])
    for k in z.files:
        if k.startswith(("score_", "noise_", "traj_", "params_", "rot_node_mask_")) or k == "num_graphs":
            continue
        setattr(d, k, torch.from_numpy(z[k]))
    d.rot_node_mask = [torch.from_numpy(z[f"rot_node_mask_{g}"]) for g in range(G)]
    d.num_graphs = G
    return d, z


def namespace_to(d, device):
    o = SimpleNamespace()
    for k, v in vars(d).items():
        if torch.is_tensor(v):
            setattr(o, k, v.to(device))
        elif isinstance(v, list):
            setattr(o, k, [x.to(device) if torch.is_tensor(x) else x for x in v])
        else:
            setattr(o, k, v)
    return o


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def oracle_batch_from_packed(raw_records, poses, pb):
    """Reference-format collated batch (oracle.pose_init.collate) whose graphs start from the very poses the device
    produced in ``pb`` (after ``assemble`` + ``init_poses``).  ``raw_records``: the per-complex record dicts the
    ComplexRecords were built from, ``poses``: per-complex pose counts; graphs complex-major like ``assemble``."""
    from diffbindfr_amd import assemble
    from oracle import pose_init as opi
    lp, ap, rp = pb.lig_ptr_host.tolist(), pb.t["atm_ptr"].cpu().tolist(), pb.res_ptr_host.tolist()
    lig_pos, rec_pos, tors = pb.lig_pos.cpu(), pb.rec_pos.cpu(), pb.torsion_angle.cpu()
    out, g = [], 0
    reps = [poses] * len(raw_records) if isinstance(poses, int) else list(poses)
    for rec, n in zip(raw_records, reps):
        fixed = assemble.ComplexRecord(copy.deepcopy(rec))
        m = fixed.atom14_mask
        feat = torch.as_tensor(rec["pocket_node_feature"]).float()
        for _ in range(n):
            pose = dict(rec)
            pose["lig_pos"] = lig_pos[lp[g]:lp[g + 1]].clone()
            pose["rec_atm_pos"] = rec_pos[ap[g]:ap[g + 1]].clone()
            pose["torsion_angle"] = tors[rp[g]:rp[g + 1]].clone()
            pose["pocket_node_feature"] = feat[m] if feat.dim() == 3 else feat
            pose["default_frame"], pose["rigid_group_positions"] = fixed.default_frame, fixed.rigid_group_positions
            pose["sc_torsion_edge_mask"], pose["atom14_mask"] = fixed.sc_mask, fixed.atom14_mask
            out.append(pose)
            g += 1
    d = SimpleNamespace(**opi.collate(out))
    d.batch = d.lig_node_batch
    return d
