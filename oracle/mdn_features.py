"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the pocket half of the MDN scorer's featurisation (row f4, next step):
`get_protein_feature` of DiffBindFR/scoring/dataset/protein_feature.py:137-216 from the point where the structure is arrays
(residue types + atom14 coordinates with absent atoms at the origin, i.e. openfold's `atom14_gt_positions`), including the
backbone part of `atom37_to_torsion_angles` (openfold/data/data_transforms.py:943-1090, vendored by the reference).

Quirks kept: the first residue's pre-omega / phi are computed against a zero "previous residue" and NOT masked; psi's
(sin, cos) is negated; chain breaks are ignored (array order); the centre of mass divides the sum over the PRESENT atoms by
the IDEAL atom count of the residue type; glycine's missing CB (slot 4) is the origin; every distance carries the reference's
`+ 1e-6` inside the norm; `cedist` is float64 (scipy distance_matrix) before the final float32 cast.
The k-nearest-neighbour graph is torch_cluster.knn_graph (absent offline, restated: the k = 30 nearest OTHER residues by CA
distance, edges j -> i grouped by i) => parity unpinned at that call; the rest is pinned by
tests/golden/make_golden.py::golden_mdn_features (tests/golden/mdn_features.npz)."""
import math

import numpy as np
import torch


def _normalize(t, dim=-1):
    return torch.nan_to_num(torch.div(t, torch.norm(t, dim=dim, keepdim=True)))


def _dihedral_sin_cos(a0, a1, a2, a3):
    """Rigid.from_3_points(a1, a2, a0).invert().apply(a3) -> (z, y) / sqrt(z^2 + y^2 + 1e-8) (rigid_utils.py:1166-1208)."""
    e0 = a2 - a1
    e1 = a0 - a2
    e0 = e0 / torch.sqrt((e0 * e0).sum(-1, keepdim=True) + 1e-8)
    dot = (e0 * e1).sum(-1, keepdim=True)
    e1 = e1 - e0 * dot
    e1 = e1 / torch.sqrt((e1 * e1).sum(-1, keepdim=True) + 1e-8)
    e2 = torch.cross(e0, e1, dim=-1)
    d = a3 - a2
    sc = torch.stack([(e2 * d).sum(-1), (e1 * d).sum(-1)], -1)
    return sc / torch.sqrt((sc * sc).sum(-1, keepdim=True) + 1e-8)


def knn_graph(ca, k):
    """torch_cluster.knn_graph(x, k) semantics: for every node i its k nearest other nodes j, edge (j -> i), grouped by i."""
    n = ca.shape[0]
    D = ((ca[:, None] - ca[None]) ** 2).sum(-1)
    D = D + torch.diag(torch.full((n,), float("inf")))
    idx = torch.argsort(D, dim=1, stable=True)[:, :min(k, n - 1)]
    return torch.stack([idx.reshape(-1), torch.arange(n).repeat_interleave(idx.shape[1])])


def pocket_features(aatype, atom14_pos, ideal_atom_count, topk=30, edge_index=None):
    """aatype [N] (< 20), atom14_pos [N,14,3] float32 (absent atoms = 0), ideal_atom_count [N] = restype_atom14_mask[aatype].sum(-1).
    Returns dict(seq, node_s [N,9], node_v [N,3,3], edge_index [2,E], edge_s [E,21], edge_v [E,1,3], xyz_full [N,14,3])."""
    X = atom14_pos.float()
    N_, CA, C, O, CB = X[:, 0], X[:, 1], X[:, 2], X[:, 3], X[:, 4]
    prev = torch.cat([torch.zeros(1, 14, 3), X[:-1]], 0)
    pre_omega = _dihedral_sin_cos(prev[:, 1], prev[:, 2], N_, CA)
    phi = _dihedral_sin_cos(prev[:, 2], N_, CA, C)
    psi = -_dihedral_sin_cos(N_, CA, C, O)
    intra = torch.stack([0.1 * torch.linalg.norm((CA - O) + 1e-6, dim=-1), 0.1 * torch.linalg.norm((N_ - O) + 1e-6, dim=-1),
                         0.1 * torch.linalg.norm((N_ - C) + 1e-6, dim=-1)]).T
    node_s = torch.cat([intra, torch.stack([pre_omega, phi, psi], 1).reshape(-1, 6)], -1)
    com = X.sum(-2) / ideal_atom_count.float()[:, None]
    if edge_index is None:
        edge_index = knn_graph(CA, topk)
    s, t = edge_index
    dmm = torch.stack([0.1 * torch.linalg.norm((CA[s] - CA[t]) + 1e-6, dim=-1), 0.1 * torch.linalg.norm((CB[s] - CB[t]) + 1e-6, dim=-1)]).T
    cd = com.double()
    cedist = (torch.sqrt(((cd[s] - cd[t]) ** 2).sum(-1)) * 0.1).view(-1, 1)
    cadist = (torch.pairwise_distance(CA[s], CA[t]) * 0.1).view(-1, 1)
    connect = (dmm[:, 0] < 4.5).float().view(-1, 1)
    mu = torch.linspace(0.0, 20.0, 16).view(1, -1)
    rbf = torch.exp(-((dmm[:, 0:1] - mu) / (20.0 / 16)) ** 2)
    edge_s = torch.cat([connect, cadist, cedist, dmm, rbf], 1)                       # float64 through cedist, cast below
    fwd = torch.nn.functional.pad(_normalize(CA[1:] - CA[:-1]), [0, 0, 0, 1])
    bwd = torch.nn.functional.pad(_normalize(CA[:-1] - CA[1:]), [0, 0, 1, 0])
    c, n = _normalize(C - CA), _normalize(N_ - CA)
    side = -_normalize(c + n) * math.sqrt(1 / 3) - _normalize(torch.cross(c, n, dim=-1)) * math.sqrt(2 / 3)
    node_v = torch.stack([fwd, bwd, side], -2)
    edge_v = _normalize(CA[s] - CA[t]).unsqueeze(-2)
    node_s, node_v, edge_s, edge_v = map(torch.nan_to_num, (node_s, node_v, edge_s, edge_v))
    return dict(seq=aatype, node_s=node_s.float(), node_v=node_v.float(), edge_index=edge_index, edge_s=edge_s.float(),
                edge_v=edge_v.float(), xyz_full=X)
