"""Row f3 measurement: dbfr_pose_metrics (HBM-bound: every trajectory byte read once) and the host PDB writer.

    python tools/export_bench.py [--poses 5120] [--frames 20]

Prints one JSON line: time per dbfr_pose_metrics call = two k_pose_metrics launches, ligand half + pocket half (hip events
on the launch stream), achieved GB/s against the
algorithmic bytes 12 * (n_lig + 14 n_res) per (pose, frame), and PDB text throughput (structures/s, MB/s) of the
library writer (the CPU restatement of the reference's Python writer is timed by tests/tools/pdb_writer_cpu_port.py).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from diffbindfr_amd import export as pex, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--poses", type=int, default=5120)
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--n-res", type=int, default=25)
ap.add_argument("--n-lig", type=int, default=30)
ap.add_argument("--perms", type=int, default=8)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="", help="lig | prot: time only that half of the kernel (developer switch)")
a = ap.parse_args()
T = synthetic.residue_tables()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
seq = rng.integers(0, 20, a.n_res)
tmask = T["atom14_mask"][seq]
tgt = (rng.standard_normal((a.n_res, 14, 3)) * 3).astype(np.float32) * tmask[..., None]
lig = (rng.standard_normal((a.n_lig, 3)) * 4).astype(np.float32)
perms = np.stack([np.arange(a.n_lig)] + [rng.permutation(a.n_lig) for _ in range(a.perms - 1)]).astype(np.int32)
pt = torch.randn(a.poses, a.frames, a.n_res, 14, 3, device=dev)
lt = torch.randn(a.poses, a.frames, a.n_lig, 3, device=dev)
args = (lt, pt, np.zeros(3, np.float32), lig, tgt, tmask, seq)
pex.pose_metrics(*args, perms=perms)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# device-resident constant inputs so that the timed region is the launch alone
import ctypes as C
from diffbindfr_amd import lib as L
lib = L.load()
f32 = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
i32 = lambda x: torch.as_tensor(x, dtype=torch.int32, device=dev).contiguous()
lg, tg, tm, aa, pm = f32(lig), f32(tgt), f32(tmask), i32(seq), i32(perms)
out = [torch.empty(a.poses, a.frames, device=dev) for _ in range(3)] + [torch.empty(a.poses, a.frames, 4, device=dev)]
cin = L.PoseMetricsIn(a.poses, a.frames, a.n_lig, a.n_res, lt.data_ptr(), pt.data_ptr(), lg.data_ptr(), tg.data_ptr(), tm.data_ptr(),
                      aa.data_ptr(), a.perms, pm.data_ptr(), None, (C.c_float * 3)(0, 0, 0), float(pex.CHI_UPPER_BOUND))
cout = L.PoseMetricsOut(out[0].data_ptr() if a.only != "prot" else None, out[1].data_ptr() if a.only != "lig" else None,
                        out[3].data_ptr() if a.only != "lig" else None, None, out[2].data_ptr() if a.only != "prot" else None)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
e0.record()
for _ in range(a.reps):
    L.check(lib.dbfr_pose_metrics(C.byref(cin), C.byref(cout), st))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
byts = 12.0 * (a.n_lig + 14 * a.n_res) * a.poses * a.frames
res = dict(kernel="k_pose_metrics", poses=a.poses, frames=a.frames, n_res=a.n_res, n_lig=a.n_lig, perms=a.perms, ms_per_call=ms,
           algorithmic_bytes=byts, achieved_GBs=byts / ms / 1e6, hbm_peak_GBs=8000.0, frac=byts / ms / 1e6 / 8000.0)

# ---- PDB writer: a 281-residue protein (the 3DBS fixture), 40 poses
z = np.load(os.path.join(ROOT, "tests", "golden", "export.npz"))
topo = pex.ProteinTopology(z["aatype"], z["atom37_pos"], z["atom37_mask"], z["residue_index"], z["chain_index"], z["b_factors"],
                           str(z["remark"]), np.nonzero(z["pocket_mask"])[0])
pose = (z["prot_traj"][:, -1] + z["center"]).astype(np.float32)
poses40 = np.concatenate([pose] * 10)
with tempfile.TemporaryDirectory() as d:
    paths = [os.path.join(d, f"p{i}.pdb") for i in range(40)]
    for th in (1, 0):
        t0 = time.perf_counter()
        topo.write_poses(poses40, paths, threads=th)
        dt = time.perf_counter() - t0
        size = os.path.getsize(paths[0])
        res[f"pdb_write_40_poses_s_threads_{th or os.cpu_count()}"] = dt
    res["pdb_bytes_per_structure"] = size
    res["pdb_structures_per_s"] = 40 / dt
    res["pdb_MBs"] = 40 * size / dt / 1e6
print(json.dumps(res))
