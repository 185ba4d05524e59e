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


def automorphisms(node_labels, edge_index, edge_labels=None, limit=100000):
    """All label-preserving automorphisms of the ligand graph, as int arrays ``perm`` with ``perm[a]`` = the atom that
    takes the place of atom ``a`` -- the list ``match_graphs(nxg, nxg, keep_self=True)`` yields for ``symm_rmsd``
    (druglib/utils/torch_utils/isom_graph.py:81-137, DiffBindFR/metrics/lrmsd.py:287-310), identity included.
    Plain backtracking over atoms in breadth-first order with (label, degree) pruning; raises if more than ``limit``
    are found (the reference bounds the same search by a 600 s timeout and then falls back to the identity).
    Returns int32 [n_perm, n_atoms]."""
    labels = np.asarray(node_labels)
    n = labels.shape[0]
    ei = np.asarray(edge_index)
    elab = {}
    adj = [set() for _ in range(n)]
    for k, (u, v) in enumerate(ei.T.tolist()):
        if u == v:
            continue
        adj[u].add(v)
        adj[v].add(u)
        if edge_labels is not None and u < v:                      # to_nx keeps the u < v direction's attribute
            elab[(u, v)] = elab[(v, u)] = np.asarray(edge_labels[k]).tolist()
    deg = [len(a) for a in adj]
    key = [(np.asarray(labels[i]).tolist(), deg[i]) for i in range(n)]
    # visiting order: breadth-first from atom 0 (every later atom has an earlier neighbour inside its component)
    order, seen = [], [False] * n
    for s in range(n):
        if seen[s]:
            continue
        seen[s] = True
        queue = [s]
        while queue:
            a = queue.pop(0)
            order.append(a)
            for b in sorted(adj[a]):
                if not seen[b]:
                    seen[b] = True
                    queue.append(b)
    earlier = [[b for b in order[:k] if b in adj[a]] for k, a in enumerate(order)]
    image = [-1] * n
    used = [False] * n
    out = []

    def place(k):
        if k == n:
            out.append(list(image))
            if len(out) > limit:
                raise ValueError(f"more than {limit} automorphisms")
            return
        a = order[k]
        anchors = earlier[k]
        cands = adj[image[anchors[0]]] if anchors else range(n)
        for c in cands:
            if used[c] or key[c] != key[a]:
                continue
            ok = True
            for b in anchors:
                if image[b] not in adj[c] or (elab and elab[(a, b)] != elab[(c, image[b])]):
                    ok = False
                    break
            if not ok:
                continue
            # non-edges among placed atoms are preserved automatically: degrees match and every edge maps to an edge
            image[a], used[c] = c, True
            place(k + 1)
            image[a], used[c] = -1, False

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, n + 200))
    try:
        place(0)
    finally:
        sys.setrecursionlimit(old)
    return np.asarray(out, np.int32).reshape(len(out), n)
