"""Reference batch dict -> packed device buffers of the C ABI (include/dbfr.h: dbfr_batch).

Input is the collated dict the reference sampler receives (SURVEY.md Appendix B.1;
produced by druglib/data/collate.py + druglib/datasets/Docking/formatting.py:6-26):
batch vectors, int64 indices, a python list of per-ligand ``rot_node_mask``.  Output:
CSR pointers, int32 indices, the ragged masks flattened -- everything a kernel needs
with coalesced, index-arithmetic-free access.  Torch is used for device memory only.
"""
import ctypes as C

import torch

from . import lib as L


def _get(data, key):
    return data[key] if isinstance(data, dict) else getattr(data, key)


def _rot_masks(data):
    """list[G] of bool [n_tor_g, N_l_g]; the reference keeps it in data.metastore (scFlex.py:137)."""
    for holder in (data, _get(data, "metastore") if _has(data, "metastore") else None):
        if holder is not None and _has(holder, "rot_node_mask"):
            return _get(holder, "rot_node_mask")
    raise KeyError("rot_node_mask")


def _has(data, key):
    return key in data if isinstance(data, dict) else hasattr(data, key)


def _ptr_from_batch(batch, G):
    cnt = torch.bincount(batch, minlength=G)
    ptr = torch.zeros(G + 1, dtype=torch.int64, device=batch.device)
    ptr[1:] = torch.cumsum(cnt, 0)
    return ptr


class PackedBatch:
    """Owns the device tensors and the ctypes ``dbfr_batch`` view onto them."""

    def __init__(self, data, device):
        dev = torch.device(device)
        g = lambda k: _get(data, k)
        to = lambda t, dt=None: (t.to(dev) if dt is None else t.to(device=dev, dtype=dt)).contiguous()
        i32 = torch.int32
        lig_batch = to(g("lig_node_batch"), torch.int64)
        atm_batch = to(g("rec_atm_pos_batch"), torch.int64)
        G = int(lig_batch.max().item()) + 1
        self.G = G
        T = {}
        lig_ptr = _ptr_from_batch(lig_batch, G)
        atm_ptr = _ptr_from_batch(atm_batch, G)
        T["lig_ptr"], T["atm_ptr"] = lig_ptr.to(i32), atm_ptr.to(i32)
        T["lig_node"] = to(g("lig_node"), torch.float32)
        T["lig_pos"] = to(g("lig_pos"), torch.float32).clone()
        NL = T["lig_pos"].shape[0]
        # ---- ligand bonds, CSR by source atom (the reference sorts by src*N+dst: ligand.py:568-570)
        ei = to(g("lig_edge_index"), torch.int64)
        EB = ei.shape[1]
        perm = torch.argsort(ei[0], stable=True)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(EB, device=dev)
        T["bond_src"], T["bond_dst"] = ei[0][perm].to(i32), ei[1][perm].to(i32)
        T["bond_feat"] = to(g("lig_edge_feat"), torch.float32)[perm].contiguous()
        bptr = torch.zeros(NL + 1, dtype=torch.int64, device=dev)
        bptr[1:] = torch.cumsum(torch.bincount(ei[0], minlength=NL), 0)
        T["bond_ptr"] = bptr.to(i32)
        # ---- ligand torsions in masked-edge order (= order of tor_score)
        tmask = to(g("tor_edge_mask"), torch.bool)
        tor_edges = torch.nonzero(tmask).flatten()
        NTOR = int(tor_edges.numel())
        T["tor_bond"] = inv[tor_edges].to(i32) if NTOR else torch.zeros(1, dtype=i32, device=dev)
        tor_graph = lig_batch[ei[0][tor_edges]]
        T["tor_ptr"] = _ptr_from_batch(tor_graph, G).to(i32)
        rot_list = _rot_masks(data)
        rows, offs, off = [], [], 0
        for m in rot_list:
            m = torch.as_tensor(m).to(torch.uint8)
            for r in range(m.shape[0]):
                offs.append(off)
                off += m.shape[1]
            rows.append(m.reshape(-1))
        assert len(offs) == NTOR, "rot_node_mask rows must match tor_edge_mask.sum()"
        T["rot_mask"] = (torch.cat(rows) if rows and off else torch.zeros(1, dtype=torch.uint8)).to(dev).contiguous()
        T["rot_mask_off"] = torch.tensor(offs if offs else [0], dtype=torch.int64, device=dev)
        # ---- pocket
        T["pocket_feat"] = to(g("pocket_node_feature"), torch.float32)
        T["rec_pos"] = to(g("rec_atm_pos"), torch.float32).clone()
        NA = T["rec_pos"].shape[0]
        T["sequence"] = to(g("sequence"), i32)
        NR = T["sequence"].shape[0]
        T["backbone_transl"] = to(g("backbone_transl"), torch.float32)
        T["backbone_rots"] = to(g("backbone_rots"), torch.float32)
        T["default_frame"] = to(g("default_frame"), torch.float32)
        T["rigid_group_positions"] = to(g("rigid_group_positions"), torch.float32)
        T["torsion_angle"] = to(g("torsion_angle"), torch.float32).clone()
        m14 = to(g("atom14_mask"), torch.bool)
        slot = torch.cumsum(m14.reshape(-1).to(torch.int64), 0) - 1
        slot = torch.where(m14.reshape(-1), slot, torch.full_like(slot, -1)).reshape(NR, 14)
        assert int(m14.sum()) == NA, "atom14_mask.sum() must equal the number of pocket atoms"
        T["atom14_slot"] = slot.to(i32)
        first = torch.where(m14, slot, torch.full_like(slot, NA)).min(dim=1).values.clamp(max=NA - 1)
        res_batch = atm_batch[first]
        T["res_ptr"] = _ptr_from_batch(res_batch, G).to(i32)
        scm = to(g("sc_torsion_edge_mask"), torch.bool)
        sc_idx = torch.nonzero(scm.reshape(-1)).flatten()
        NSC = int(sc_idx.numel())
        T["sc_res_chi"] = sc_idx.to(i32) if NSC else torch.zeros(1, dtype=i32, device=dev)
        tei = to(g("torsion_edge_index"), torch.int64).reshape(NR * 4, 2)
        T["sc_bond"] = tei[sc_idx].to(i32).contiguous() if NSC else torch.zeros(1, 2, dtype=i32, device=dev)
        T["sc_ptr"] = _ptr_from_batch(res_batch[sc_idx // 4], G).to(i32)
        self._finish(T, scm, m14, lig_ptr, atm_ptr, T["res_ptr"])

    @classmethod
    def from_tensors(cls, T, sc_mask, atom14_mask):
        """A PackedBatch over already packed device tensors (diffbindfr_amd.assemble builds them directly
        from per-complex records, without the per-pose dicts and the collate)."""
        self = cls.__new__(cls)
        self.G = int(T["lig_ptr"].numel()) - 1
        self._finish(T, sc_mask, atom14_mask, T["lig_ptr"].long(), T["atm_ptr"].long(), T["res_ptr"])
        return self

    def _finish(self, T, scm, m14, lig_ptr, atm_ptr, res_ptr):
        self.t = T
        self.sc_mask = scm
        self.atom14_mask = m14
        self.lig_ptr_host = lig_ptr.cpu()
        self.res_ptr_host = res_ptr.cpu().long()
        apc = atm_ptr.cpu()
        dims = dict(G=self.G, NL=int(T["lig_pos"].shape[0]), NA=int(T["rec_pos"].shape[0]), NR=int(T["sequence"].shape[0]),
                    EB=int(T["bond_src"].shape[0]), NTOR=int(self.tor_count(T)), NSC=int(scm.sum().item()),
                    max_nl=int((self.lig_ptr_host[1:] - self.lig_ptr_host[:-1]).max().item()),
                    max_na=int((apc[1:] - apc[:-1]).max().item()),
                    max_nr=int((self.res_ptr_host[1:] - self.res_ptr_host[:-1]).max().item()))
        self.dims = dims
        self.c = L.Batch()
        for k, v in dims.items():
            setattr(self.c, k, v)
        for k in L._BATCH_PTRS:
            setattr(self.c, k, C.c_void_p(T[k].data_ptr()))

    @staticmethod
    def tor_count(T):
        return int(T["tor_ptr"][-1].item())

    # convenience views on the evolving state
    @property
    def lig_pos(self):
        return self.t["lig_pos"]

    @property
    def rec_pos(self):
        return self.t["rec_pos"]

    @property
    def torsion_angle(self):
        return self.t["torsion_angle"]
