"""Output side of the sampler: per-pose metrics where the trajectories are (the device) and the
PDB text of every pose on host threads -- the work ``complex_modeling`` does per pose in Python
(DiffBindFR/evaluation/export.py:106-312: ``calc_lig_centroid`` / ``chi_differ`` / ``sidechain_rmsd`` at :139-195,
``prot_final.pdb`` / ``pkt_final.pdb`` at :261-274; ``symm_rmsd`` DiffBindFR/metrics/lrmsd.py:287-335).

Not covered (toolkit work outside the numeric path): the RDKit SDF writer and RDKit's own symmetry RMSD
(``calc_rmsd``), the MDAnalysis XTC writer, smina.  No CPU path for the metrics: a CPU tensor raises ``DbfrError``.
"""
import ctypes as C
import datetime
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import lib as L

CHI_UPPER_BOUND = 15 / 180 * np.pi            # export.py:123


def pose_metrics(lig_traj, prot_traj, center, lig_target, atom14_target, atom14_target_mask, aatype, perms=None,
                 heavy_mask=None, chi_bound=CHI_UPPER_BOUND, with_delta_chi=False):
    """One ``dbfr_pose_metrics`` launch over the trajectories of one complex.

    lig_traj [P,T,N_l,3], prot_traj [P,T,N_r,14,3] (pocket-centred, as ``DiffBindFRHIP`` returns them), center [3] the
    pocket centre, lig_target [N_l,3] absolute, atom14_target [N_r,14,3] pocket-centred, atom14_target_mask [N_r,14],
    aatype [N_r]; perms int [n_perm,N_l] from ``ligand.automorphisms`` (None: identity only).
    Returns dict(centroid [P,T], sc_rmsd [P,T], chi_rate [P,T,4], lig_rmsd [P,T] (+ delta_chi [P,T,N_r,4])) on the device.
    """
    lib = L.load()
    dev = prot_traj.device
    if dev.type != "cuda" or lig_traj.device != dev:
        raise L.DbfrError("pose_metrics needs ROCm device tensors (no CPU path)")
    P, Tn, n_lig = (int(x) for x in lig_traj.shape[:3])
    n_res = int(prot_traj.shape[2])
    if tuple(prot_traj.shape) != (P, Tn, n_res, 14, 3) or tuple(lig_traj.shape) != (P, Tn, n_lig, 3):
        raise L.DbfrError(f"trajectory shapes {tuple(lig_traj.shape)} / {tuple(prot_traj.shape)}")
    f32 = lambda x: torch.as_tensor(x).to(device=dev, dtype=torch.float32).contiguous()
    i32 = lambda x: torch.as_tensor(x).to(device=dev, dtype=torch.int32).contiguous()
    lt, pt, lg, tg, tm, aa = f32(lig_traj), f32(prot_traj), f32(lig_target), f32(atom14_target), f32(atom14_target_mask), i32(aatype)
    if pt.data_ptr() % 8:
        pt = pt.clone()
    pm = i32(np.arange(n_lig, dtype=np.int32)[None] if perms is None else perms)
    if pm.dim() != 2 or pm.shape[1] != n_lig:
        raise L.DbfrError("perms must be [n_perm, n_lig]")
    hm = None if heavy_mask is None else i32(np.asarray(heavy_mask).astype(np.int32))
    out = dict(centroid=torch.empty(P, Tn, device=dev), sc_rmsd=torch.empty(P, Tn, device=dev),
               chi_rate=torch.empty(P, Tn, 4, device=dev), lig_rmsd=torch.empty(P, Tn, device=dev))
    if with_delta_chi:
        out["delta_chi"] = torch.empty(P, Tn, n_res, 4, device=dev)
    p = lambda x: None if x is None else x.data_ptr()
    c = [float(v) for v in torch.as_tensor(center).reshape(3).tolist()]
    cin = L.PoseMetricsIn(P, Tn, n_lig, n_res, p(lt), p(pt), p(lg), p(tg), p(tm), p(aa), int(pm.shape[0]), p(pm), p(hm),
                          (C.c_float * 3)(*c), float(chi_bound))
    cout = L.PoseMetricsOut(p(out["centroid"]), p(out["sc_rmsd"]), p(out["chi_rate"]), p(out.get("delta_chi")), p(out["lig_rmsd"]))
    L.check(lib.dbfr_pose_metrics(C.byref(cin), C.byref(cout), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out          # temporaries above are released in stream order by torch's caching allocator (same stream as the launch)


@dataclass
class ProteinTopology:
    """The static part of a structure file: what ``Protein`` (druglib/utils/obj/protein.py:37-91) holds that ``to_pdb``
    reads.  ``pocket_rows``: indices of the pocket residues inside the protein (``Protein.pocket_mask``)."""
    aatype: np.ndarray            # [N] 0..20
    atom37_pos: np.ndarray        # [N,37,3]
    atom37_mask: np.ndarray       # [N,37]
    residue_index: np.ndarray     # [N]
    chain_index: np.ndarray       # [N]
    b_factors: np.ndarray         # [N,37]
    remark: Optional[str] = None
    pocket_rows: Optional[np.ndarray] = None

    def __post_init__(self):
        self.aatype = np.ascontiguousarray(self.aatype, np.int32)
        n = self.aatype.shape[0]
        self.atom37_pos = np.ascontiguousarray(self.atom37_pos, np.float32).reshape(n, 37, 3)
        self.atom37_mask = np.ascontiguousarray(self.atom37_mask, np.float32).reshape(n, 37)
        self.residue_index = np.ascontiguousarray(self.residue_index, np.int32).reshape(n)
        self.chain_index = np.ascontiguousarray(self.chain_index, np.int32).reshape(n)
        self.b_factors = np.ascontiguousarray(self.b_factors, np.float64).reshape(n, 37)
        if self.pocket_rows is not None:
            self.pocket_rows = np.ascontiguousarray(self.pocket_rows, np.int32)

    def pocket(self):
        """The pocket as its own structure (``pkt_final.pdb``)."""
        r = self.pocket_rows
        return ProteinTopology(self.aatype[r], self.atom37_pos[r], self.atom37_mask[r], self.residue_index[r],
                               self.chain_index[r], self.b_factors[r], self.remark)

    def _c(self, remark):
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        return L.PdbTopology(int(self.aatype.shape[0]), ptr(self.aatype), ptr(self.atom37_pos), ptr(self.atom37_mask),
                             ptr(self.residue_index), ptr(self.chain_index), ptr(self.b_factors),
                             None if remark is None else remark.encode())

    def _remark(self, version):
        if self.remark is not None:
            return self.remark
        return 'REMARK   1 CREATED WITH MDLDruglib %s, %s' % (version, str(datetime.date.today()))     # protein.py:711

    def _rows(self, pos14, rows):
        if pos14 is None:
            return 0, None, None
        a = np.ascontiguousarray(torch.as_tensor(pos14).detach().cpu().numpy() if isinstance(pos14, torch.Tensor) else pos14, np.float32)
        if rows is None and a.shape[-3] != self.aatype.shape[0]:
            rows = self.pocket_rows
        r = None if rows is None else np.ascontiguousarray(rows, np.int32)
        return (a.shape[-3] if r is None else r.shape[0]), r, a

    def to_pdb(self, pos14=None, rows=None, model=None, add_end=True, version="1.0.0"):
        """``Protein.pos_update(pos14).to_pdb()``: pos14 [n_rows,14,3] replaces the atom14 coordinates of ``rows``
        (default: all residues if it covers them, else the pocket rows)."""
        lib = L.load()
        n_rows, r, a = self._rows(pos14, rows)
        if a is not None and a.shape != (n_rows, 14, 3):
            raise L.DbfrError(f"pos14 shape {a.shape}")
        topo = self._c(self._remark(version) if (model is None or model == 1 or self.remark is not None) else None)
        pr = None if r is None else r.ctypes.data_as(C.c_void_p)
        pa = None if a is None else a.ctypes.data_as(C.c_void_p)
        need = lib.dbfr_pdb_format(C.byref(topo), n_rows, pr, pa, -1 if model is None else int(model), int(add_end), None, 0)
        if need < 0:
            L.check(int(need))
        buf = C.create_string_buffer(int(need))
        got = lib.dbfr_pdb_format(C.byref(topo), n_rows, pr, pa, -1 if model is None else int(model), int(add_end), buf, need)
        assert got == need
        return buf.raw.decode()

    def write_poses(self, pos14, paths, rows=None, threads=0, version="1.0.0"):
        """One file per pose: pos14 [n_pose, n_rows, 14, 3] -> paths[i] (``prot_final.pdb`` of every sample) on host
        threads inside the library."""
        lib = L.load()
        n_rows, r, a = self._rows(pos14, rows)
        if a.ndim != 4 or a.shape[1:] != (n_rows, 14, 3) or a.shape[0] != len(paths):
            raise L.DbfrError(f"pos14 shape {a.shape} for {len(paths)} paths")
        topo = self._c(self._remark(version))
        arr = (C.c_char_p * len(paths))(*[str(x).encode() for x in paths])
        L.check(lib.dbfr_pdb_write_files(C.byref(topo), n_rows, None if r is None else r.ctypes.data_as(C.c_void_p),
                                         a.ctypes.data_as(C.c_void_p), len(paths), arr, int(threads)))
