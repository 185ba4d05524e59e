"""Ligand topology on the host: rotatable bonds and the atoms they move.

Replaces ``find_torsion`` / ``TorsionFactory`` (druglib/datasets/Docking/utils.py:47-92, mol_pipeline.py:90-110): a directed
bond (u -> v) is a torsion iff removing it disconnects the molecule, the smaller fragment has more than one atom and
contains v; its ``rot_node_mask`` row is that fragment.  Once per ligand, pure graph code (no chemistry toolkit).
"""
import numpy as np


def _component(adj, n, start, skip_u, skip_v):
    seen = np.zeros(n, bool)
    seen[start] = True
    stack = [start]
    while stack:
        a = stack.pop()
        for b in adj[a]:
            if (a == skip_u and b == skip_v) or (a == skip_v and b == skip_u):
                continue
            if not seen[b]:
                seen[b] = True
                stack.append(b)
    return seen


def torsion_masks(n_atoms, edge_index):
    """edge_index [2,E] directed covalent bonds (both directions present).  Returns (tor_edge_mask bool [E],
    rot_node_mask bool [n_tor, n_atoms]) in bond order, like the reference."""
    ei = np.asarray(edge_index)
    adj = [[] for _ in range(n_atoms)]
    for u, v in ei.T.tolist():
        adj[u].append(v)
    tor = np.zeros(ei.shape[1], bool)
    rows = []
    for k, (u, v) in enumerate(ei.T.tolist()):
        side_v = _component(adj, n_atoms, v, u, v)
        if side_v[u]:
            continue                                  # still connected: a ring bond
        nv = int(side_v.sum())
        if nv * 2 == n_atoms:
            small = side_v if side_v[0] else ~side_v  # equal halves: the fragment that holds atom 0 (first component networkx lists)
        else:
            small = side_v if nv * 2 < n_atoms else ~side_v
        if int(small.sum()) > 1 and small[v]:
            tor[k] = True
            rows.append(small.copy())
    rot = np.asarray(rows, bool) if rows else np.zeros((0, n_atoms), bool)
    return tor, rot
