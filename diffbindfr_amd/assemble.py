"""Batch assembly straight from per-complex records.

The reference prepares every one of the ``complexes x poses`` graphs separately on the host:
``InferenceDataset._prepare_test_sample`` deep-copies the per-complex model input, runs the
real-time transforms ``LigInit -> SCFixer -> SCProtInit -> Atom14ToAllAtomsRepr``
(DiffBindFR/common/inference_dataset.py:578-612, druglib/datasets/Docking/struct_init.py,
formatting.py:41-51) and the DataLoader collates the per-pose dicts
(druglib/data/collate.py:18-137, formatting.py:6-26).  With the sampler on the device that
Python becomes the bottleneck, so here

  * ``ComplexRecord``  does the once-per-complex work on the host (SCFixer, bond sort,
    atom14 compaction) -- O(complexes), not O(poses);
  * ``assemble``       tiles the static per-complex arrays into the packed CSR batch of
    ``include/dbfr.h`` with a handful of device gathers (no per-pose Python, no collate);
  * ``init_poses``     randomises every pose on the device (``dbfr_init_poses``: LigInit +
    SCProtInit + side-chain rebuild) from a tape of draws made with the torch generator.

Record keys are the reference's collected keys (DiffBindFR/configs/diffbindfr_ts.py:49-55)
plus ``rot_node_mask`` (``metastore['rot_node_mask']``).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import lib as L
from .tables import residue_tables
from .packing import PackedBatch


def _t(x, dtype=None):
    x = torch.as_tensor(np.asarray(x)) if not torch.is_tensor(x) else x
    return x if dtype is None else x.to(dtype)


class _Half:
    """One half (ligand or pocket) of a complex: host tensors + a per-device cache of their uploads, so that a ligand
    or pocket shared by many complexes (forward screen: one receptor; target fishing: one ligand) and a complex
    sampled in several batches cross the PCIe link ONCE."""
    _FIELDS = ()

    def dev(self, device):
        device = torch.device(device)
        cache = self.__dict__.setdefault("_dev", {})
        if device not in cache:
            cache[device] = {k: getattr(self, k).to(device) for k in self._FIELDS}
        return cache[device]

    def release(self):
        self.__dict__.pop("_dev", None)

    def nbytes(self):
        """Bytes of one device copy of this half."""
        return sum(getattr(self, k).numel() * getattr(self, k).element_size() for k in self._FIELDS)


class LigandRecord(_Half):
    """Static host-side form of one ligand (mol_pipeline.py: LigandFeaturizer / TorsionFactory / LigandGrapher keys)."""
    _FIELDS = ("lig_node", "lig_pos", "bond_src", "bond_dst", "bond_feat", "bond_deg", "tor_bond", "rot_mask")

    def __init__(self, rec):
        g = lambda k: rec[k] if isinstance(rec, dict) else getattr(rec, k)
        meta = g("metastore") if (isinstance(rec, dict) and "metastore" in rec) or hasattr(rec, "metastore") else None
        rot = meta["rot_node_mask"] if meta is not None and "rot_node_mask" in meta else g("rot_node_mask")
        self.lig_node = _t(g("lig_node"), torch.float32)
        self.lig_pos = _t(g("lig_pos"), torch.float32)
        n_l = self.n_l = self.lig_pos.shape[0]
        ei = _t(g("lig_edge_index"), torch.int64)
        perm = torch.argsort(ei[0], stable=True)          # CSR by source atom, the reference's order inside a row
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel())
        self.bond_src, self.bond_dst = ei[0][perm], ei[1][perm]
        self.bond_feat = _t(g("lig_edge_feat"), torch.float32)[perm]
        self.bond_deg = torch.bincount(ei[0], minlength=n_l)
        tmask = _t(g("tor_edge_mask")).bool()
        self.tor_bond = inv[torch.nonzero(tmask).flatten()]            # masked bonds in the reference's order
        self.n_tor = int(self.tor_bond.numel())
        rot = _t(rot).to(torch.uint8).reshape(self.n_tor, n_l)
        self.rot_mask = rot.reshape(-1)


class PocketRecord(_Half):
    """Static host-side form of one pocket (pocket_pipeline.py keys), SCFixer applied."""
    _FIELDS = ("sequence", "sc_mask", "atom14_mask", "default_frame", "rigid_group_positions", "backbone_transl",
               "backbone_rots", "torsion_angle", "pocket_feat", "atom14_slot", "sc_res_chi", "sc_bond")

    def __init__(self, rec, tables=None):
        tables = tables or residue_tables()
        g = lambda k: rec[k] if isinstance(rec, dict) else getattr(rec, k)
        # SCFixer first: it may change masks and templates
        seq = _t(g("sequence"), torch.int64)
        scm = _t(g("sc_torsion_edge_mask")).bool().clone()
        m14 = _t(g("atom14_mask")).bool().clone()
        df = _t(g("default_frame"), torch.float32).clone()
        rg = _t(g("rigid_group_positions"), torch.float32).clone()
        scm, m14, df, rg = sc_fixer(seq, scm, m14, df, rg, tables)
        self.sequence, self.sc_mask, self.atom14_mask = seq, scm, m14
        self.default_frame, self.rigid_group_positions = df, rg
        self.n_r = seq.shape[0]
        self.n_a = int(m14.sum())
        self.backbone_transl = _t(g("backbone_transl"), torch.float32)
        self.backbone_rots = _t(g("backbone_rots"), torch.float32)
        self.torsion_angle = _t(g("torsion_angle"), torch.float32)
        feat = _t(g("pocket_node_feature"), torch.float32)
        self.pocket_feat = feat[m14] if feat.dim() == 3 else feat       # atom14 layout or already compacted
        assert self.pocket_feat.shape[0] == self.n_a, "pocket_node_feature rows must match atom14_mask.sum()"
        # CA / CB atoms (atom37 ids 1 and 3, tpscore.py:630-640): the complete-bipartite part of the cross graph
        self.n_cab = int(((self.pocket_feat[:, 0] == 1) | (self.pocket_feat[:, 0] == 3)).sum())
        slot = torch.cumsum(m14.reshape(-1).long(), 0) - 1
        self.atom14_slot = torch.where(m14.reshape(-1), slot, torch.full_like(slot, -1)).reshape(self.n_r, 14)
        sc_idx = torch.nonzero(scm.reshape(-1)).flatten()
        self.sc_res_chi = sc_idx
        self.n_sc = int(sc_idx.numel())
        tei = _t(g("torsion_edge_index"), torch.int64).reshape(self.n_r * 4, 2)
        self.sc_bond = tei[sc_idx]


class ComplexRecord:
    """One (pocket, ligand) job: everything that does not change between its poses.  Built from a reference-format
    per-complex dict, or from two halves -- complexes that share a ``LigandRecord`` / ``PocketRecord`` OBJECT share its
    device upload (``assemble`` detects the sharing by identity)."""

    def __init__(self, rec=None, tables=None, lig=None, pocket=None):
        self.lig = lig if lig is not None else LigandRecord(rec)
        self.pocket = pocket if pocket is not None else PocketRecord(rec, tables)

    def __getattr__(self, k):           # read access to the halves' fields (n_l, bond_src, sequence, ...)
        if k in ("lig", "pocket"):
            raise AttributeError(k)
        for h in (self.lig, self.pocket):
            if k in h.__dict__:
                return h.__dict__[k]
        raise AttributeError(k)

    @property
    def cost(self):
        """Edge-count cost of one pose-step."""
        from .dist import complex_cost
        return complex_cost(self.n_a, self.n_l, self.n_cab)


def sc_fixer(seq, sc_mask, atom14_mask, default_frame, rigid_group_positions, tables):
    """SCFixer (struct_init.py:72-106): residues whose chi mask disagrees with their type fall back to the AF2
    literature frames, positions and atom14 mask -- provided every residue of the pocket has its N, CA, C
    (the reference tests the backbone of the whole pocket at once, :83)."""
    exists = torch.from_numpy(np.asarray(tables["chi_mask"])).bool()[seq]
    bad = (exists != sc_mask).any(dim=-1)
    if not bool(bad.any()):
        return sc_mask, atom14_mask, default_frame, rigid_group_positions
    bb_ok = bool(atom14_mask[:, :3].all())
    if not bb_ok:
        return torch.zeros_like(sc_mask), atom14_mask, default_frame, rigid_group_positions
    res = seq[bad]
    sc_mask[bad] = exists[bad]
    default_frame[bad] = torch.from_numpy(np.asarray(tables["default_frame"])).float()[res]
    rigid_group_positions[bad] = torch.from_numpy(np.asarray(tables["atom14_lit_pos"])).float()[res]
    atom14_mask[bad] = torch.from_numpy(np.asarray(tables["atom14_mask"])).bool()[res]
    return sc_mask, atom14_mask, default_frame, rigid_group_positions


def _tile(counts_c, cg, dev):
    """Ragged replication: complex c owns ``counts_c[c]`` consecutive rows of a concatenated array; graph g is a copy
    of complex ``cg[g]``.  Returns (src row of every batch row, graph of every batch row, ptr [G+1])."""
    counts_c = counts_c.to(dev)
    base_c = torch.cumsum(counts_c, 0) - counts_c
    cnt_g = counts_c[cg]
    ptr = torch.zeros(cg.numel() + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(cnt_g, 0)
    total = int(ptr[-1].item())
    graph = torch.repeat_interleave(torch.arange(cg.numel(), device=dev), cnt_g, output_size=total)
    src = base_c[cg][graph] + (torch.arange(total, device=dev) - ptr[graph])
    return src, graph, ptr


def _uniq(objs):
    """(unique objects in first-seen order, index of every object into that list) -- sharing is by identity."""
    seen, uniq, idx = {}, [], []
    for o in objs:
        if id(o) not in seen:
            seen[id(o)] = len(uniq)
            uniq.append(o)
        idx.append(seen[id(o)])
    return uniq, torch.as_tensor(idx, dtype=torch.int64)


def assemble(records, poses, device):
    """Packed batch of ``sum(poses)`` graphs; graphs of one complex are consecutive (complex-major), like the
    reference's dataset order (InferenceDataset repeats every pair ``num_poses`` times).  ``lig_pos`` holds the input
    conformers, ``torsion_angle`` the input psi/chi and ``rec_pos`` zeros until ``init_poses`` runs.
    Static per-ligand / per-pocket arrays are read from the halves' device caches (uploaded once per device) and
    replicated per pose by device gathers; a half shared by several complexes is concatenated once."""
    dev = torch.device(device)
    n_c = len(records)
    reps = torch.as_tensor([poses] * n_c if isinstance(poses, int) else list(poses), dtype=torch.int64)
    assert reps.numel() == n_c and int(reps.min()) > 0
    ligs, lig_of_c = _uniq([r.lig for r in records])
    pkts, pkt_of_c = _uniq([r.pocket for r in records])
    cg = torch.repeat_interleave(torch.arange(n_c), reps)
    lg, pg = lig_of_c[cg].to(dev), pkt_of_c[cg].to(dev)     # ligand / pocket of every graph
    cg = cg.to(dev)
    dl, dp = [h.dev(dev) for h in ligs], [h.dev(dev) for h in pkts]
    catl = lambda k: torch.cat([d[k] for d in dl], 0)
    catp = lambda k: torch.cat([d[k] for d in dp], 0)
    cntl = lambda f: torch.as_tensor([f(r) for r in ligs], dtype=torch.int64)
    cntp = lambda f: torch.as_tensor([f(r) for r in pkts], dtype=torch.int64)
    i32 = torch.int32
    T = {}
    # ---------------- ligand atoms
    a_src, _, lig_ptr = _tile(cntl(lambda r: r.n_l), lg, dev)
    T["lig_ptr"] = lig_ptr.to(i32)
    T["lig_node"] = catl("lig_node")[a_src].contiguous()
    T["lig_pos"] = catl("lig_pos")[a_src].contiguous()
    deg = catl("bond_deg")[a_src]
    bptr = torch.zeros(a_src.numel() + 1, dtype=torch.int64, device=dev)
    bptr[1:] = torch.cumsum(deg, 0)
    T["bond_ptr"] = bptr.to(i32)
    # ---------------- bonds (already sorted by source atom inside each ligand)
    b_src, b_graph, bond_gptr = _tile(cntl(lambda r: r.bond_src.numel()), lg, dev)
    off = lig_ptr[b_graph]
    T["bond_src"] = (catl("bond_src")[b_src] + off).to(i32)
    T["bond_dst"] = (catl("bond_dst")[b_src] + off).to(i32)
    T["bond_feat"] = catl("bond_feat")[b_src].contiguous()
    # ---------------- ligand torsions + their rotating-side masks
    t_src, t_graph, tor_ptr = _tile(cntl(lambda r: r.n_tor), lg, dev)
    T["tor_ptr"] = tor_ptr.to(i32)
    n_tor = int(tor_ptr[-1].item())
    if n_tor:
        T["tor_bond"] = (catl("tor_bond")[t_src] + bond_gptr[t_graph]).to(i32)
        m_src, _, mask_ptr = _tile(cntl(lambda r: r.n_tor * r.n_l), lg, dev)
        T["rot_mask"] = catl("rot_mask")[m_src].contiguous()
        nl_g = (lig_ptr[1:] - lig_ptr[:-1])
        T["rot_mask_off"] = (mask_ptr[t_graph] + (torch.arange(n_tor, device=dev) - tor_ptr[t_graph]) * nl_g[t_graph]).contiguous()
    else:
        T["tor_bond"] = torch.zeros(1, dtype=i32, device=dev)
        T["rot_mask"] = torch.zeros(1, dtype=torch.uint8, device=dev)
        T["rot_mask_off"] = torch.zeros(1, dtype=torch.int64, device=dev)
    # ---------------- pocket atoms / residues
    p_src, _, atm_ptr = _tile(cntp(lambda r: r.n_a), pg, dev)
    T["atm_ptr"] = atm_ptr.to(i32)
    T["pocket_feat"] = catp("pocket_feat")[p_src].contiguous()
    T["rec_pos"] = torch.zeros(p_src.numel(), 3, device=dev)
    r_src, r_graph, res_ptr = _tile(cntp(lambda r: r.n_r), pg, dev)
    T["res_ptr"] = res_ptr.to(i32)
    T["sequence"] = catp("sequence")[r_src].to(i32)
    for k in ("backbone_transl", "backbone_rots", "default_frame", "rigid_group_positions", "torsion_angle"):
        T[k] = catp(k)[r_src].contiguous()
    slot = catp("atom14_slot")[r_src]
    T["atom14_slot"] = torch.where(slot >= 0, slot + atm_ptr[r_graph][:, None], slot).to(i32).contiguous()
    m14 = catp("atom14_mask")[r_src]
    scm = catp("sc_mask")[r_src]
    # ---------------- side-chain torsions
    s_src, s_graph, sc_ptr = _tile(cntp(lambda r: r.n_sc), pg, dev)
    T["sc_ptr"] = sc_ptr.to(i32)
    if s_src.numel():
        T["sc_res_chi"] = (catp("sc_res_chi")[s_src] + 4 * res_ptr[s_graph]).to(i32)
        T["sc_bond"] = (catp("sc_bond")[s_src] + atm_ptr[s_graph][:, None]).to(i32).contiguous()
    else:
        T["sc_res_chi"] = torch.zeros(1, dtype=i32, device=dev)
        T["sc_bond"] = torch.zeros(1, 2, dtype=i32, device=dev)
    pb = PackedBatch.from_tensors(T, scm, m14)
    pb.complex_of_graph = cg
    return pb


def draw_init_tape_dims(G, n_tor, n_res, dev, tr_sigma_max=10.0, generator=None):
    """The per-pose draws of LigInit / SCProtInit (struct_init.py:30-33,41-48,118-120) for ``G`` graphs with ``n_tor``
    ligand torsions and ``n_res`` residues in total, made on ``dev`` with the torch generator: U(-pi,pi) torsion kicks,
    uniformly random rotations (normalised N(0,1) quaternions, as scipy's Rotation.random), N(0, tr_sigma_max)
    translations, U(-pi,pi) chi draws."""
    u = lambda *shape: (torch.rand(*shape, device=dev, generator=generator) * 2.0 - 1.0) * math.pi
    q = torch.randn(G, 4, device=dev, generator=generator)
    q = q / q.norm(dim=-1, keepdim=True)
    x, y, z, w = q.unbind(-1)                          # scipy's scalar-last convention
    rot = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                       2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(G, 3, 3)
    return dict(tor=u(n_tor), rot=rot.contiguous(), tr=torch.randn(G, 3, device=dev, generator=generator) * tr_sigma_max,
                sc=u(n_res, 4))


def draw_init_tape(pb, tr_sigma_max=10.0, generator=None):
    """``draw_init_tape_dims`` for a whole packed batch from ONE generator."""
    d = pb.dims
    return draw_init_tape_dims(d["G"], max(d["NTOR"], 1), d["NR"], pb.lig_pos.device, tr_sigma_max, generator)


@torch.no_grad()
def init_poses(model, pb, tape, atom14_out=None):
    """Randomise every pose of ``pb`` in place on the device (``dbfr_init_poses``).  ``model`` is the
    TensorProductModelHIP that owns the library handle; ``tape`` as ``draw_init_tape`` returns it."""
    lib = L.load()
    dev = pb.lig_pos.device
    if dev.type != "cuda":
        raise L.DbfrError("init_poses needs a ROCm device (no CPU path)")
    f = lambda k: tape[k].to(device=dev, dtype=torch.float32).contiguous()
    keep = {k: f(k) for k in ("tor", "rot", "tr", "sc")}
    z = L.InitTape(*(C.c_void_p(keep[k].data_ptr()) for k in ("tor", "rot", "tr", "sc")))
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = C.c_void_p(atom14_out.data_ptr()) if atom14_out is not None else None
    with torch.cuda.device(dev):
        L.check(lib.dbfr_init_poses(model.handle(dev), C.byref(pb.c), C.byref(z), out, stream))
    pb._init_tape = keep            # keep the device tape alive until the stream has consumed it
    return pb
