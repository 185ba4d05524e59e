"""The reference's own multi-complex examples (README.md:86-127) against the reference's own numbers.

tests/golden/real_forward15_traj.npz: examples/forward = the 3DBS receptor with all 15 SDF ligands (BASELINE config 3 in
miniature: every job shares ONE pocket); tests/golden/real_reverse_traj.npz: examples/reverse = 2 ligands x 3 receptors
(config 4: jobs share ligands).  One pose per pair, all 20 steps through the REFERENCE's `DiffBindFR.sample()`
(tests/golden/make_golden.py: golden_examples; the oracle reproduced it at <= 1e-4 A there).  The fixtures hold the RAW
inputs of the product path -- protein atom37 arrays around the site, the pocket-defining crystal ligand, ligand graphs --
plus the recorded tapes.  GPU: pockets_from_proteins (selection + templates on the device) -> shared PocketRecord /
LigandRecord -> ComplexRecord -> assemble / dbfr_init_poses from the recorded init tapes -> 20 steps with the recorded
noise: every trajectory within 1e-3 A of the reference's, and the same final poses through dist.run_sharded.
"""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN

FIXTURES = ["real_forward15_traj.npz", "real_reverse_traj.npz"]
LIG_KEYS = ("lig_node", "lig_pos", "lig_edge_index", "lig_edge_feat", "tor_edge_mask", "rot_node_mask")


def _load(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated (tests/golden/make_golden.py examples)")
    return np.load(path)


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_shapes(name):
    z = _load(name)
    pairs = z["pairs"]
    n_p, n_l = int(z["n_prot"]), int(z["n_lig"])
    assert len(pairs) == (15 if "forward" in name else 6) and pairs[:, 0].max() == n_p - 1 and pairs[:, 1].max() == n_l - 1
    nl = [int(z[f"lig{i}_lig_pos"].shape[0]) for i in range(n_l)]
    assert z["traj_lig"].shape == (20, sum(nl[l] for _, l in pairs), 3)
    assert z["noise_tr"].shape == (20, len(pairs), 3)
    for g in range(len(pairs)):
        assert z[f"tape{g}_rot"].shape == (3, 3) and abs(np.linalg.det(z[f"tape{g}_rot"]) - 1) < 1e-6
    for i in range(n_p):
        assert z[f"prot{i}_atom37_pos"].shape[1:] == (37, 3) and z[f"prot{i}_aatype"].max() < 20


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_gpu_examples_follow_the_reference_trajectories(name):
    import diffbindfr_amd as dba
    from diffbindfr_amd import assemble, dist as ddist, pocket
    from oracle import score_model as sm
    dev = torch.device("cuda:0")
    z = _load(name)
    pairs = [tuple(p) for p in z["pairs"].tolist()]
    n_p, n_l = int(z["n_prot"]), int(z["n_lig"])
    # ---- product path: proteins -> pockets (ONE device pass for all receptors) -> shared halves
    aa = np.concatenate([z[f"prot{i}_aatype"] for i in range(n_p)])
    p37 = np.concatenate([z[f"prot{i}_atom37_pos"] for i in range(n_p)])
    m37 = np.concatenate([z[f"prot{i}_atom37_mask"] for i in range(n_p)])
    ref = np.concatenate([z[f"prot{i}_ref_lig_pos"] for i in range(n_p)])
    rp = np.cumsum([0] + [len(z[f"prot{i}_aatype"]) for i in range(n_p)])
    fp = np.cumsum([0] + [len(z[f"prot{i}_ref_lig_pos"]) for i in range(n_p)])
    halves, mask = pocket.pockets_from_proteins(torch.from_numpy(aa).to(dev), torch.from_numpy(p37).to(dev), torch.from_numpy(m37).to(dev),
                                                torch.from_numpy(ref).to(dev), cutoff=12.0, res_ptr=rp, ref_ptr=fp)
    if "forward" in name:
        assert int(mask.sum()) == 105                                       # the 12 A pocket of 3DBS
    pockets = [assemble.PocketRecord({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in h.items()}) for h in halves]
    ligs = [assemble.LigandRecord({k: torch.from_numpy(z[f"lig{i}_{k}"]) for k in LIG_KEYS}) for i in range(n_l)]
    jobs = [assemble.ComplexRecord(lig=ligs[l], pocket=pockets[p]) for p, l in pairs]
    assert len({id(j.pocket) for j in jobs}) == n_p and len({id(j.lig) for j in jobs}) == n_l      # halves are shared, not copied
    params = sm.init_params(sm.default_cfg(), seed=int(z["params_seed"]))
    model = dba.TensorProductModelHIP({}).to(dev)
    model.load_state_dict(params, strict=True)
    samp = dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})
    # ---- recorded tapes, job by job (one pose each)
    T = 20
    tapes, to, so = {}, 0, 0
    for g, job in enumerate(jobs):
        init = dict(tor=z[f"tape{g}_tor"], rot=z[f"tape{g}_rot"][None], tr=z[f"tape{g}_tr"].reshape(1, 3), sc=z[f"tape{g}_sc"])
        zz = dict(tr=z["noise_tr"][:, g:g + 1], rot=z["noise_rot"][:, g:g + 1], tor=z["noise_tor"][:, to:to + job.n_tor],
                  sc=z["noise_sc"][:, so:so + job.n_sc])
        to += job.n_tor
        so += job.n_sc
        tapes[g] = (init, zz)
    assert to == z["noise_tor"].shape[1] and so == z["noise_sc"].shape[1]
    # ---- (a) the whole table as one batch, all 20 frames
    pb, lig, a14 = samp.run_complexes(jobs, 1, dev, seeds=[0] * len(jobs), tapes=[tapes[g] for g in range(len(jobs))], visualize=True)
    dl = (lig.cpu() - torch.from_numpy(z["traj_lig"])).norm(dim=-1)
    assert lig.shape[0] == T and float(dl.max()) < 1e-3, float(dl.max())
    da = (a14[-1].cpu() - torch.from_numpy(z["final_atom14"])).norm(dim=-1)
    assert float(da.max()) < 1e-3, float(da.max())
    # ---- (b) through the job driver in small batches: the same final poses, in job order
    res = ddist.run_sharded(samp, jobs, 1, seed=0, device=dev, batch_poses=4, tapes=tapes)
    lp, rp_ = pb.lig_ptr_host.tolist(), pb.res_ptr_host.tolist()
    for g, (l, a) in enumerate(res):
        assert (l[0].cpu() - torch.from_numpy(z["traj_lig"][-1, lp[g]:lp[g + 1]])).norm(dim=-1).max() < 1e-3
        assert (a[0].cpu() - torch.from_numpy(z["final_atom14"][rp_[g]:rp_[g + 1]])).norm(dim=-1).max() < 1e-3
