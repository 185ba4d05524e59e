"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's rotatable-bond finder
(druglib/datasets/Docking/utils.py:47-92 ``find_torsion``; networkx semantics spelled out):
for every directed covalent bond (u, v) in edge order, remove it from the undirected molecule graph; if the
graph falls apart, take the smallest connected component -- ``sorted(components, key=len)[0]``, a stable sort over
networkx's component order, i.e. components ordered by their lowest atom index -- and mark the bond as a torsion
iff that component has more than one atom and contains v.  Pinned against the reference's own function on the
3DBS crystal ligand and on synthetic molecules by tests/golden/make_golden.py.
"""
import numpy as np


def _components(n, und_edges):
    adj = [[] for _ in range(n)]
    for a, b in und_edges:
        adj[a].append(b)
        adj[b].append(a)
    seen, comps = [False] * n, []
    for s in range(n):                      # networkx iterates nodes in insertion order 0..n-1
        if seen[s]:
            continue
        comp, stack = [], [s]
        seen[s] = True
        while stack:
            a = stack.pop()
            comp.append(a)
            for b in adj[a]:
                if not seen[b]:
                    seen[b] = True
                    stack.append(b)
        comps.append(comp)
    return comps


def find_torsion(n_atoms, edge_index):
    ei = np.asarray(edge_index)
    und = sorted({(min(u, v), max(u, v)) for u, v in ei.T.tolist()})
    tor, rows = [], []
    for u, v in ei.T.tolist():
        rest = [e for e in und if e != (min(u, v), max(u, v))]
        comps = _components(n_atoms, rest)
        hit = False
        if len(comps) > 1:
            small = sorted(comps, key=len)[0]
            if len(small) > 1 and v in small:
                m = np.zeros(n_atoms, bool)
                m[np.asarray(small, int)] = True
                rows.append(m)
                hit = True
        tor.append(hit)
    return np.asarray(tor, bool), (np.asarray(rows, bool) if rows else np.zeros((0, n_atoms), bool))
