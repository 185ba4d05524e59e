"""Freeze the AF2 residue constant TABLES the hot path consumes as data.

Runs in the build container only (imports the reference's
druglib/utils/obj/protein_constants.py through tests/golden/ref_shims.py) and
writes diffbindfr_amd/data/residue_tables.npz.  Tables (numeric data, no code):
  atom14_to_group   [21,14] int   protein_constants.py:1177 restype_atom14_to_rigid_group
  atom14_mask       [21,14] f32   :1178 restype_atom14_mask
  atom14_lit_pos    [21,14,3] f32 :1179 restype_atom14_rigid_group_positions
  default_frame     [21,8,4,4]    :1180 restype_rigid_group_default_frame
  torsion_edges     [21,4,2] int  :1181,1279 restype_atom14_torsion_edges[..., 1, :] (j-k bond of chi)
  chi_mask          [21,4]        :87 chi_angles_mask (+ an all-zero row for 'X')
  atom14_to_atom37  [21,14] int   :1331 atoms37_to_atoms14_mapper (atom37 id of each atom14 slot)
  atom37_to_coarse  [37], atom37_to_element [37]   :612-614
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "tests", "golden"))
import ref_shims  # noqa: E402

ns = ref_shims.load_hot_path()
pc = ns.pc
chi = np.zeros((21, 4), np.float32)
chi[:20] = np.asarray(pc.chi_angles_mask, np.float32)[:20]
out = dict(
    atom14_to_group=pc.restype_atom14_to_rigid_group.astype(np.int32),
    atom14_mask=pc.restype_atom14_mask.astype(np.float32),
    atom14_lit_pos=pc.restype_atom14_rigid_group_positions.astype(np.float32),
    default_frame=pc.restype_rigid_group_default_frame.astype(np.float32),
    torsion_edges=pc.restype_atom14_torsion_edges[:, :, 1, :].astype(np.int32),
    chi_mask=chi,
    atom14_to_atom37=pc.atoms37_to_atoms14_mapper.astype(np.int32),
    atom37_to_coarse=np.asarray(pc.atom37_to_coarse_atom_type, np.int32),
    atom37_to_element=np.asarray(pc.atom37_to_atom_element, np.int32),
)
dst = os.path.join(here, "..", "diffbindfr_amd", "data", "residue_tables.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: v.shape for k, v in out.items()})
