"""One-off check (GPU box, ~3 min): the real 3DBS complex of BASELINE config 1 (tests/golden/real_3dbs.npz) through
records -> assemble -> device pose initialisation -> 20 denoise steps, against the CPU oracle fed the same initial pose
and noise tape.  Prints the deviation of the final poses.  Test infrastructure (imports oracle/)."""
import copy
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import diffbindfr_amd as dba  # noqa: E402
from diffbindfr_amd import assemble, pocket, synthetic  # noqa: E402
from oracle import pose_init as opi, sampler as osampler, schedule as osched, score_model as sm  # noqa: E402
from tests.test_real_complex import _ligand_half, fixture  # noqa: E402

dev = torch.device("cuda:0")
z = fixture()
half = pocket.pocket_records(z["aatype"], z["atom14_position"], z["atom14_mask"], None, dev)[0]
lig = _ligand_half(z)
lig["lig_pos"] = torch.from_numpy(z["lig_pos"]).float() - half["pocket_center_pos"]
rec = {**half, **lig}
mcfg = sm.default_cfg()
params = sm.init_params(mcfg, seed=1)
model = dba.TensorProductModelHIP({}).to(dev)
model.load_state_dict(params, strict=True)
samp = dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})
pb = assemble.assemble([assemble.ComplexRecord(copy.deepcopy(rec))], 1, dev)
gen = torch.Generator(device=dev).manual_seed(3)
tape = assemble.draw_init_tape(pb, 10.0, gen)
tape["tr"] *= 0.3                                     # keep the ligand near the pocket: edges on every step
assemble.init_poses(model, pb, tape)
torch.cuda.synchronize()
# the oracle starts from the very pose the device produced
T = synthetic.residue_tables()
m = half["atom14_mask"]
pose = dict(rec)
pose["lig_pos"] = pb.lig_pos.cpu().clone()
pose["torsion_angle"] = pb.torsion_angle.cpu().clone()
pose["rec_atm_pos"] = pb.rec_pos.cpu().clone()
pose["pocket_node_feature"] = half["pocket_node_feature"][m]
fixed = assemble.ComplexRecord(copy.deepcopy(rec))
pose["default_frame"], pose["rigid_group_positions"] = fixed.default_frame, fixed.rigid_group_positions
pose["sc_torsion_edge_mask"], pose["atom14_mask"] = fixed.sc_mask, fixed.atom14_mask
d = SimpleNamespace(**opi.collate([pose]))
d.batch = d.lig_node_batch
scfg = osched.default_sample_cfg()
n_tor, n_sc = int(d.tor_edge_mask.sum()), int(d.sc_torsion_edge_mask.sum())
noise = osampler.draw_noise(scfg.actual_steps, 1, n_tor, n_sc, seed=5)
zt = {k: getattr(noise, k).to(dev).contiguous() for k in ("tr", "rot", "tor", "sc")}
t0 = time.time()
lig_hip, a14_hip = samp.sample_packed(pb, zt)
torch.cuda.synchronize()
t_hip = time.time() - t0
t0 = time.time()
lig_ref, a14_ref = osampler.sample(params, mcfg, scfg, copy.deepcopy(d), noise, torch.from_numpy(T["atom14_to_group"]).long())
t_cpu = time.time() - t0
dl = (lig_hip[0].cpu() - lig_ref[0]).norm(dim=-1)
da = (a14_hip[0].cpu() - a14_ref[0]).norm(dim=-1)
print(f"3DBS real complex, 1 pose x {scfg.actual_steps} steps: HIP {t_hip:.2f} s, CPU oracle {t_cpu:.1f} s; "
      f"ligand RMSD {dl.pow(2).mean().sqrt():.2e} A, max {dl.max():.2e} A; pocket atom14 max {da.max():.2e} A")
