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


class SdfTemplate:
    """The constant text of a ligand's V2000 mol block, prepared once per ligand so that every pose's ``lig_final.sdf``
    (DiffBindFR/evaluation/export.py:236-244) is a coordinate substitution (``dbfr_sdf_write_files``, library threads).

    ``from_molblock`` takes the ligand's input SD record and, like the reference's loader (RDKit ``RemoveHs``: the model
    ligand holds heavy atoms only), drops explicit hydrogens: atoms renumbered, bonds to hydrogens removed, the atom lists
    of ``M  CHG`` / ``M  ISO`` / ``M  RAD`` lines remapped.  Only the FIRST record of a multi-record SD file is used (everything
    behind its ``$$$$`` is dropped); property lines that carry atom numbers in other layouts -- ``A  nnn`` / ``V  nnn`` (and the
    text line that follows an alias), ``G  `` , ``M  ALS`` / ``M  RGP`` / ``M  SAL`` / ``M  SBL`` / ``M  SPA`` / ``M  APO`` /
    ``M  AAL`` ... -- are dropped when hydrogens were removed (their numbers would be stale) and kept verbatim otherwise; data
    items (``> <tag>`` blocks) are kept.  Atom order = the order the sampler's poses use (heavy atoms in file order).  Byte
    parity with RDKit's ``SDWriter`` is not pinned offline (RDKit absent)."""

    def __init__(self, header, atom_tails, trailer):
        self.header, self.atom_tails, self.trailer = header, list(atom_tails), trailer
        self.n_atoms = len(self.atom_tails)

    @classmethod
    def from_molblock(cls, text, remove_hs=True, program="DBFR-HIP"):
        lines = text.replace("\r\n", "\n").split("\n")
        if len(lines) < 4 or "V2000" not in lines[3]:
            raise ValueError("not a V2000 mol block")
        na, nb = int(lines[3][0:3]), int(lines[3][3:6])
        atoms, bonds = lines[4:4 + na], lines[4 + na:4 + na + nb]
        rest = lines[4 + na + nb:]
        if "$$$$" in rest:                                                      # first record only
            rest = rest[:rest.index("$$$$") + 1]
        sym = [a[31:34].strip() for a in atoms]
        keep = [i for i in range(na) if not (remove_hs and sym[i] == "H")]
        ren = {old: new + 1 for new, old in enumerate(keep)}                    # 0-based old -> 1-based new
        out_bonds = []
        for b in bonds:
            i, j = int(b[0:3]) - 1, int(b[3:6]) - 1
            if i in ren and j in ren:
                out_bonds.append(f"{ren[i]:3d}{ren[j]:3d}{b[6:]}")
        renumbered = len(keep) != na
        props, skip_next, in_data = [], False, False
        for l in rest:
            if skip_next:                                                       # the text line of a dropped `A  nnn` alias
                skip_next = False
                continue
            if l.startswith(">"):
                in_data = True
            if in_data:
                props.append(l)
            elif l[:6] in ("M  CHG", "M  ISO", "M  RAD"):
                n = int(l[6:9])
                ent = [(int(l[9 + 8 * k:13 + 8 * k]) - 1, l[13 + 8 * k:17 + 8 * k]) for k in range(n)]
                ent = [(ren[a], v) for a, v in ent if a in ren]
                if ent:
                    props.append(f"{l[:6]}{len(ent):3d}" + "".join(f"{a:4d}{v}" for a, v in ent))
            elif renumbered and (l[:3] in ("A  ", "V  ", "G  ") or (l[:3] == "M  " and l[:6] != "M  END")):
                skip_next = l[:3] == "A  "                                      # atom-indexed in a layout not remapped here: stale -> dropped
            else:
                props.append(l)
        while props and props[-1] == "":
            props.pop()
        if "M  END" not in props:
            props.insert(0, "M  END")
        if not props or props[-1] != "$$$$":
            props.append("$$$$")
        counts = f"{len(keep):3d}{len(out_bonds):3d}{lines[3][6:]}"
        header = "\n".join([lines[0], f"  {program:<8s}          3D", lines[2], counts]) + "\n"
        trailer = "\n".join(out_bonds + props) + "\n"
        return cls(header, [atoms[i][30:] for i in keep], trailer)

    def _c(self):
        from . import lib as L
        import ctypes as C
        self._keep = [t.encode() for t in self.atom_tails]
        arr = (C.c_char_p * max(self.n_atoms, 1))(*self._keep)
        self._arr = arr
        return L.SdfTemplate(self.n_atoms, self.header.encode(), arr, self.trailer.encode())

    def format(self, pos):
        """Text of one pose (pos [n_atoms,3])."""
        from . import lib as L
        import ctypes as C
        lib = L.load()
        a = np.ascontiguousarray(np.asarray(pos, np.float32).reshape(self.n_atoms, 3))
        t = self._c()
        need = lib.dbfr_sdf_format(C.byref(t), a.ctypes.data_as(C.c_void_p), None, 0)
        if need < 0:
            L.check(int(need))
        buf = C.create_string_buffer(int(need))
        lib.dbfr_sdf_format(C.byref(t), a.ctypes.data_as(C.c_void_p), buf, need)
        return buf.raw.decode()

    def write_poses(self, pos, paths, threads=0):
        """pos [n_pose, n_atoms, 3] -> paths[i], on library threads."""
        from . import lib as L
        import ctypes as C
        lib = L.load()
        a = np.ascontiguousarray(np.asarray(pos, np.float32))
        if a.ndim != 3 or a.shape[1:] != (self.n_atoms, 3) or a.shape[0] != len(paths):
            raise L.DbfrError(f"pose array {a.shape} for {len(paths)} paths of a {self.n_atoms}-atom ligand")
        t = self._c()
        arr = (C.c_char_p * len(paths))(*[str(x).encode() for x in paths])
        L.check(lib.dbfr_sdf_write_files(C.byref(t), a.ctypes.data_as(C.c_void_p), len(paths), arr, int(threads)))
