"""CPU baseline for the PDB writer of row f3: the restated reference writer (oracle/export.pose_pdb, kind "port") timed on
the 3DBS fixture structure (2 279 atoms), next to tools/export_bench.py's numbers for the library writer.

    python tests/tools/pdb_writer_cpu_port.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import export as oex  # noqa: E402
from diffbindfr_amd import tables  # noqa: E402

T = tables.residue_tables()
z = np.load(os.path.join(ROOT, "tests", "golden", "export.npz"))
pose = (z["prot_traj"][:, -1] + z["center"]).astype(np.float32)
rows = np.nonzero(z["pocket_mask"])[0]
t0 = time.perf_counter()
for i in range(4):
    txt = oex.pose_pdb(z["aatype"], z["atom37_pos"], z["atom37_mask"], z["residue_index"], z["chain_index"], z["b_factors"], rows, pose[i], T,
                       str(z["remark"]))
dt = time.perf_counter() - t0
print(json.dumps({"cpu_port_pdb_structures_per_s": 4 / dt, "bytes_per_structure": len(txt), "cores": 1, "kind": "port",
                  "sample": "4 poses of the 281-residue 3DBS fixture protein"}))
