"""Developer timing of the scorer (row f4): P poses of one complex of the 3DBS shape through KarmaDockHIP.score_poses
(device pocket features + ligand encoder once + GVP encoder + mixture head); poses/s by HIP-event timing."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from diffbindfr_amd import mdn  # noqa: E402
from tests.test_mdn_inputs import mdn_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--poses", type=int, default=640)
ap.add_argument("--n-res", type=int, default=105)
ap.add_argument("--n-lig", type=int, default=35)
a = ap.parse_args()
dev = torch.device("cuda:0")
m = mdn.KarmaDockHIP().to(dev)
g = torch.Generator().manual_seed(0)
sd = m.state_dict()
for k, v in sd.items():
    if v.dim() == 2:
        v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) / np.sqrt(v.shape[1]))
m.load_state_dict(sd, strict=True)
rng = np.random.default_rng(0)
d = mdn_inputs(rng, [(a.n_lig, a.n_res)], coincident=False)
lig = {k: d[k] for k in ("lig_node_s", "lig_edge_s", "lig_edge_index")}
aa = d["pro_seq"].clamp(max=19)
x = d["pro_xyz_full"]
poses = (x[None] + 0.3 * torch.randn(a.poses, *x.shape)).to(dev)
lpos = (d["lig_pos"][None] + 0.5 * torch.randn(a.poses, a.n_lig, 3)).to(dev)
m.score_poses(lig, aa, poses, lpos)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    s = m.score_poses(lig, aa, poses, lpos)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = min(ts)
print(f"scorer: {a.poses} poses x ({a.n_res} residues, {a.n_lig} ligand atoms): {t:.2f} ms = {a.poses / t * 1e3:.0f} poses/s "
      f"(all {', '.join(f'{v:.1f}' for v in ts)}); score[0] {float(s[0]):.4f}")
