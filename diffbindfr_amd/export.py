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
    with torch.cuda.device(dev):
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


@dataclass
class ComplexOutput:
    """What ``complex_modeling`` reads per complex from the reference's dataset objects
    (``PD.traj_group[name]``, ``proteins[key]``, ``ligands[key]``; export.py:125-137,150-159,198-203)."""
    name: str
    ligand_traj: torch.Tensor                 # [N_pose, N_traj, N_l, 3] pocket-centred (device)
    protein_traj: torch.Tensor                # [N_pose, N_traj, N_r, 14, 3] pocket-centred (device)
    pocket_center_pos: np.ndarray             # proteinmeta.pocket_center_pos
    ligand_pos: np.ndarray                    # ligandmeta.ligand.atom_positions (absolute)
    ligand_labels: np.ndarray                 # atom types / elements (automorphisms must preserve them)
    ligand_edge_index: np.ndarray             # [2,E] covalent bonds, both directions
    topology: ProteinTopology                 # proteinmeta.protein with pocket_rows = nonzero(pocket_mask)
    atom14_position: np.ndarray               # proteinmeta.atom14_position (pocket-centred), [N_r,14,3]
    atom14_mask: np.ndarray                   # proteinmeta.atom14_mask
    aatype: np.ndarray                        # proteinmeta.pocket.aatype
    row: Optional[dict] = None                # the pair_frame row copied onto every pose (export.py:143-148)
    heavy_mask: Optional[np.ndarray] = None
    sdf_template: Optional[object] = None     # ligand.SdfTemplate of the input SD record -> lig_final.sdf per pose


def rmsd_to_str(rmsd):
    """export.py:32-36."""
    return str(round(rmsd, 2)).replace('.', '_')


def complex_modeling(entries, export_dir=None, calc_metrics=False, lrmsd_naming=False, complex_name_split=None,
                     ligand_writer=None, threads=0, **kwargs):
    """``complex_modeling`` (DiffBindFR/evaluation/export.py:106-312) over ``ComplexOutput`` entries: same flags
    (``export_fullp``, ``export_pkt``), same directory layout (``<export_dir>/<name>/sample_<i>[_<rmsd>]/prot_final.pdb`` /
    ``pkt_final.pdb``), same frame columns (``centroid``, ``chi1_15``, ``sc-rmsd``, ``l-rmsd``, ``sample_id``, ``docked_lig``,
    ``protein_pdb``) and the same ``arr_df`` dict.  Differences: the metrics of all poses come from one device launch per complex;
    ``l-rmsd`` is the heavy-atom RMSD minimised over the graph automorphisms (the reference asks RDKit for the symmetry
    classes); ``lig_final.sdf`` is written from the entry's ``sdf_template`` (``ligand.SdfTemplate``: the input mol block with
    the pose's coordinates, library threads) or by ``ligand_writer(entry, pose_index, final_pos[N_l,3], path)`` if given (RDKit's
    SDWriter in the reference); the trajectory flags (``export_fullp_traj`` / ``export_pkt_traj``: ligand HETATM records + XTC)
    are not supported."""
    import pandas as pd
    from collections import defaultdict
    from . import ligand as _ligand
    if kwargs.get("export_fullp_traj") or kwargs.get("export_pkt_traj"):
        raise NotImplementedError("trajectory export (PLComplex.to_pdb + MDAnalysis XTC) stays with the reference")
    fullp, pkt = bool(kwargs.get("export_fullp", False)), bool(kwargs.get("export_pkt", False))
    df, pd_df = defaultdict(list), defaultdict(list)
    for e in entries:
        n_pose = int(e.ligand_traj.shape[0])
        center = torch.as_tensor(np.asarray(e.pocket_center_pos), dtype=torch.float32).reshape(3)
        for k, v in (e.row or {}).items():
            pd_df[k].extend([v] * n_pose)
        lrmsd = None
        if calc_metrics:
            try:
                perms = _ligand.automorphisms(e.ligand_labels, e.ligand_edge_index)
            except ValueError as err:       # highly symmetric ligand: like the reference after its matcher timeout
                import warnings             # (metrics/lrmsd.py:299-309: identity mapping only), not an aborted export
                warnings.warn(f"{e.name}: {err}; l-rmsd without symmetry correction")
                perms = None
            m = pose_metrics(e.ligand_traj, e.protein_traj, center, e.ligand_pos, e.atom14_position, e.atom14_mask, e.aatype,
                             perms=perms, heavy_mask=e.heavy_mask)
            last = {k: m[k][:, -1].cpu() for k in ("centroid", "sc_rmsd", "lig_rmsd")}
            chi1 = m["chi_rate"][:, -1, 0].cpu()
            for col, val in (("centroid", last["centroid"]), ("chi1_15", chi1), ("sc-rmsd", last["sc_rmsd"])):
                df[col].append(val.tolist())
                pd_df[col].extend(val.tolist())
            lrmsd = last["lig_rmsd"].tolist()
        if not (fullp or pkt) or export_dir is None:
            continue                                       # like the reference (:198): no 'l-rmsd' / 'sample_id' columns without export
        if calc_metrics:
            pd_df["l-rmsd"].extend(lrmsd)
            df["l-rmsd"].append(lrmsd)
        import os
        name = e.name.split(complex_name_split)[-1] if complex_name_split is not None else e.name
        compl_dir = os.path.join(str(export_dir), name)
        ids = [f"sample_{i + 1}" + (f"_{rmsd_to_str(lrmsd[i])}" if (calc_metrics and lrmsd_naming) else "") for i in range(n_pose)]
        dirs = [os.path.join(compl_dir, s) for s in ids]
        for d in dirs:
            os.makedirs(d, exist_ok=True)
        final_prot = (e.protein_traj[:, -1] + center.to(e.protein_traj.device)).cpu()        # add_center_pos, export.py:136
        if fullp:
            e.topology.write_poses(final_prot, [os.path.join(d, "prot_final.pdb") for d in dirs], threads=threads)
        if pkt:
            e.topology.pocket().write_poses(final_prot, [os.path.join(d, "pkt_final.pdb") for d in dirs], threads=threads)
        want_lig = ligand_writer is not None or e.sdf_template is not None
        final_lig = (e.ligand_traj[:, -1] + center.to(e.ligand_traj.device)).cpu().numpy() if want_lig else None
        if ligand_writer is None and e.sdf_template is not None:      # all poses of the complex in one library call
            e.sdf_template.write_poses(final_lig, [os.path.join(d, "lig_final.sdf") for d in dirs], threads=threads)
        for i, d in enumerate(dirs):
            pd_df["sample_id"].append(ids[i])
            sdf = os.path.join(d, "lig_final.sdf")
            if ligand_writer is not None:
                ligand_writer(e, i, final_lig[i], sdf)
            pd_df["docked_lig"].append(sdf)
            pd_df["protein_pdb"].append(os.path.join(d, "prot_final.pdb" if fullp else "pkt_final.pdb"))
    return pd.DataFrame(pd_df), ({k: np.array(v) for k, v in df.items()} if calc_metrics else None)
