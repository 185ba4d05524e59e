"""Featurised synthetic inputs of the MDN scorer (test infrastructure shared by tests/test_mdn.py and the golden
generator): per graph (n_lig, n_res) with the reference's feature widths (DiffBindFR/scoring/dataset/ligand_feature.py,
protein_feature.py: 89 + 20 ligand, 9 + 3x3 residue, 21 + 1x3 pocket edge, 30 nearest CA per residue)."""
import numpy as np
import torch

from diffbindfr_amd import synthetic


def mdn_inputs(rng, sizes, coincident=True):
    d = {k: [] for k in ("lig_node_s", "lig_edge_s", "lig_edge_index", "lig_pos", "lig_batch", "pro_node_s", "pro_node_v",
                         "pro_edge_index", "pro_edge_s", "pro_edge_v", "pro_seq", "pro_xyz_full", "pro_batch")}
    lo = po = 0
    for b, (nl, nr) in enumerate(sizes):
        lig = synthetic.make_ligand(rng, nl)
        ei = lig["lig_edge_index"]
        ca = rng.normal(0, 6.0, (nr, 3))
        xyz = ca[:, None, :] + rng.normal(0, 1.5, (nr, 14, 3))
        xyz[:, 1] = ca
        absent = rng.random((nr, 14)) < 0.25
        absent[:, :5] = False
        xyz[absent] = 0.0                                             # unused atom14 slots sit at the origin (not masked by the head)
        lig_pos = ca[rng.integers(0, nr, nl)] + rng.normal(0, 2.5, (nl, 3))
        if coincident:
            lig_pos[0] = xyz[0, 2]                                     # one coincident pair: d^2 rounds to <= 0
        k = min(30, nr - 1)
        D = np.linalg.norm(ca[:, None] - ca[None], axis=-1) + np.eye(nr) * 1e9
        nbr = np.argsort(D, axis=1)[:, :k]
        pei = np.stack([nbr.reshape(-1), np.repeat(np.arange(nr), k)])   # knn_graph: source = neighbour, target = centre
        ev = ca[pei[0]] - ca[pei[1]]
        ev = ev / np.linalg.norm(ev, axis=-1, keepdims=True)
        d["lig_node_s"].append(rng.normal(0, 1, (nl, 89))); d["lig_edge_s"].append(rng.normal(0, 1, (ei.shape[1], 20)))
        d["lig_edge_index"].append(ei + lo); d["lig_pos"].append(lig_pos); d["lig_batch"].append(np.full(nl, b))
        d["pro_node_s"].append(rng.normal(0, 1, (nr, 9))); d["pro_node_v"].append(rng.normal(0, 1, (nr, 3, 3)))
        d["pro_edge_index"].append(pei + po); d["pro_edge_s"].append(rng.normal(0, 1, (pei.shape[1], 21)))
        d["pro_edge_v"].append(ev[:, None, :]); d["pro_seq"].append(rng.integers(0, 21, nr)); d["pro_xyz_full"].append(xyz)
        d["pro_batch"].append(np.full(nr, b))
        lo += nl; po += nr
    out = {}
    for k, v in d.items():
        a = np.concatenate(v, axis=1 if k.endswith("edge_index") else 0)
        out[k] = torch.from_numpy(a).long() if a.dtype.kind == "i" else torch.from_numpy(a).float()
    return out
