"""How sensitive is the reference ALGORITHM (not this implementation) to rounding-sized input changes on its own examples?

tests/golden/real_forward15_traj.npz / real_reverse_traj.npz hold the reference's 20-step trajectories of examples/forward (15 poses)
and examples/reverse (6 poses).  The HIP sampler follows all but one trajectory per fixture to < 1e-4 A; one departs by 0.01-0.04 A
(tests/test_examples.py).  This script runs the CPU ORACLE (max|d| = 0 against the reference on these very batches, make_golden.py)
twice on the same batch: (A) as frozen -- must reproduce the fixture -- and (B) with the initial ligand / pocket coordinates moved by
N(0, 1e-6 A), the size of fp32 rounding at these coordinates.  Per-pose deviation B - A after 20 steps shows whether hard neighbour
cutoffs (4 A / 5 A / 0.2 sigma + 5 A, no envelope on the edge features) make single trajectories jump under such a change.

    python tests/tools/example_sensitivity.py [forward|reverse] [sigma]     (about 2 x 2 min on 128 host threads)
"""
import copy
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from diffbindfr_amd import synthetic  # noqa: E402
from oracle import pose_init as opi, sampler, schedule, score_model as sm  # noqa: E402

HALF_KEYS = ("sequence", "atom14_mask", "backbone_transl", "backbone_rots", "default_frame", "rigid_group_positions", "torsion_angle",
             "torsion_edge_index", "sc_torsion_edge_mask", "pocket_node_feature")
LIG_KEYS = ("lig_node", "lig_pos", "lig_edge_index", "lig_edge_feat", "tor_edge_mask", "rot_node_mask")

which = sys.argv[1] if len(sys.argv) > 1 else "forward"
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
z = np.load(os.path.join(ROOT, "tests", "golden", "real_forward15_traj.npz" if which == "forward" else "real_reverse_traj.npz"))
T = synthetic.residue_tables()
Tt = {k: (torch.from_numpy(np.asarray(v)) if k == "atom14_to_group" else v) for k, v in T.items()}
poses = []
for g, (p, l) in enumerate(z["pairs"].tolist()):
    rec = {k: torch.from_numpy(z[f"prot{p}_half_{k}"]) for k in HALF_KEYS}
    rec.update({k: torch.from_numpy(z[f"lig{l}_{k}"]) for k in LIG_KEYS})
    tape = dict(tor=z[f"tape{g}_tor"], rot=z[f"tape{g}_rot"], tr=torch.from_numpy(z[f"tape{g}_tr"]).float(), sc=z[f"tape{g}_sc"])
    poses.append(opi.init_pose(opi.sc_fixer(copy.deepcopy(rec), T), tape, Tt))
d = types.SimpleNamespace(**opi.collate(poses))
d.batch = d.lig_node_batch
G = d.num_graphs
mcfg = sm.default_cfg()
params = sm.init_params(mcfg, seed=int(z["params_seed"]))
scfg = schedule.default_sample_cfg()
noise = types.SimpleNamespace(**{k: torch.from_numpy(z[f"noise_{k}"]) for k in ("tr", "rot", "tor", "sc")})
a14g = torch.from_numpy(T["atom14_to_group"]).long()


def run(dd):
    t0 = time.time()
    lig, a14 = sampler.sample(params, mcfg, scfg, copy.deepcopy(dd), noise, a14g, torus_seed=0, visualize=True)
    return lig, a14, time.time() - t0


ligA, a14A, dt = run(d)
ref = torch.from_numpy(z["traj_lig"])
print(f"{which}: {G} poses, oracle run A in {dt:.0f}s on {torch.get_num_threads()} threads; vs the frozen reference trajectory: max|d| = {float((ligA - ref).abs().max()):.2e} A")
gen = torch.Generator().manual_seed(1)
dB = copy.deepcopy(d)
dB.lig_pos = d.lig_pos + sigma * torch.randn(d.lig_pos.shape, generator=gen)
ligB, a14B, dt = run(dB)
lp = d.lig_node_ptr.tolist()
dev = (ligB - ligA).norm(dim=-1)
per = [[float(dev[s, lp[g]:lp[g + 1]].max()) for s in (0, 4, 9, 14, 19)] for g in range(G)]
print(f"run B = run A with the initial ligand coordinates moved by N(0, {sigma:g} A): per pose max deviation (A) after steps 1, 5, 10, 15, 20")
for g, row in enumerate(per):
    print(f"  pose {g:2d}: " + "  ".join(f"{x:.2e}" for x in row) + ("   <-- departs" if row[-1] > 1e-3 else ""))
print(f"poses within 1e-3 A after 20 steps: {sum(r[-1] < 1e-3 for r in per)} of {G}; largest {max(r[-1] for r in per):.3e} A")
