"""The oracle against the golden fixtures frozen from the REFERENCE's own source
(tests/golden/make_golden.py, run in the build container)."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import geometry, sampler, schedule, score_model as sm
from diffbindfr_amd import synthetic
from tests.helpers import GOLDEN, load_golden_batch

T = synthetic.residue_tables()


def t(x):
    return torch.from_numpy(np.asarray(x))


def test_geometry_fixture():
    z = np.load(os.path.join(GOLDEN, "geometry.npz"))
    assert (geometry.axis_angle_to_rot(t(z["aa"])) - t(z["aa_rot"])).abs().max() < 1e-6
    R, tr = geometry.kabsch(t(z["kabsch_A"]), t(z["kabsch_B"]))
    assert (R - t(z["kabsch_R"])).abs().max() < 1e-5 and (tr - t(z["kabsch_t"])).abs().max() < 1e-5
    G = int(z["lig_lig_node_batch"].max()) + 1
    rot = [t(z[f"lig_rot_node_mask_{g}"]) for g in range(G)]
    new = geometry.update_batchlig_pos(t(z["lig_tr"]), t(z["lig_rot"]), t(z["lig_tor"]), t(z["lig_lig_pos"]),
                                       t(z["lig_lig_edge_index"]), t(z["lig_tor_edge_mask"]), rot, t(z["lig_lig_node_batch"]))
    assert (new - t(z["lig_new_pos"])).abs().max() < 2e-5
    a14 = geometry.build_atom14(t(z["sc_seq"]), t(z["sc_transl"]), t(z["sc_rots"]), t(z["sc_default_frame"]),
                                t(z["sc_rigid"]), t(z["sc_angle"]), t(T["atom14_to_group"]).long())
    assert (a14 - t(z["sc_atom14"])).abs().max() < 2e-5


def test_embedding_fixture():
    z = np.load(os.path.join(GOLDEN, "embeddings.npz"))
    assert torch.equal(sm.sinusoidal_embedding(1000 * t(z["temb_t"]), 32), t(z["temb"]))
    for stop in (4, 5, 32):
        off = torch.linspace(0.0, float(stop), 32)
        p = {"x.offset": off, "x.coeff": -0.5 / (off[1] - off[0]) ** 2}
        assert (sm.gaussian_smearing(p, "x", t(z[f"gs{stop}_d"])) - t(z[f"gs{stop}"])).abs().max() < 1e-7
    assert torch.equal(sm.complete_bipartite(torch.tensor([2, 3, 2]), torch.tensor([4, 2, 5])), t(z["bip"]))
    p = {"n.mean_shift": t(z["ln_mean_shift"]), "n.affine_weight": t(z["ln_weight"]), "n.affine_bias": t(z["ln_bias"])}
    y = sm.layer_norm(p, "n", "48x0e + 12x1o + 12x1e + 48x0o", t(z["ln_x"]))
    assert (y - t(z["ln_y"])).abs().max() < 1e-6


def test_schedule_fixture():
    z = np.load(os.path.join(GOLDEN, "schedule.npz"))
    cfg = schedule.default_sample_cfg()
    for i, row in enumerate(z["steps"]):
        sc = schedule.step_scalars(cfg, i)
        got = [float(sc.t), float(sc.dt), float(sc.tr_sigma), float(sc.rot_sigma), float(sc.tor_sigma),
               float(sc.sc_tor_sigma), float(sc.tr_g), float(sc.rot_g), float(sc.tor_g)]
        assert np.allclose(got, row, rtol=0, atol=0)
    for i, v in zip(z["so3_idx"], z["so3_val"]):
        assert abs(schedule.so3_exp_score_norm(int(i)) - v) <= 1e-12 * max(1.0, abs(v))
    for i, v in zip(z["torus_idx"], z["torus_norm_seed0"]):
        got = schedule.torus_score_norm_entry(int(i), 0)
        assert got == v or (np.isnan(got) and np.isnan(v))
    # SURVEY.md Appendix B.4 spot values
    assert abs(z["steps"][0][2] - 6.0) < 1e-5 and abs(z["steps"][19][2] - 0.17478) < 1e-4
    assert list(schedule.so3_eps_index(np.array([1.55, 0.25799, 0.05138], np.float32))) == [952, 613, 309]


def _params_digest(params):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(k.encode())
        h.update(params[k].numpy().tobytes())
    return h.hexdigest()


def test_score_model_and_sampler_fixture():
    d, z = load_golden_batch()
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=int(z["params_seed"]))
    assert _params_digest(params) == str(z["params_sha256"]), "seeded parameters differ from the fixture's"
    scfg = schedule.default_sample_cfg()
    G = d.num_graphs
    for step in (0, 19):
        sc = schedule.step_scalars(scfg, step)
        out = sm.forward(params, mcfg, sampler.set_time(copy.deepcopy(d), sc, G))
        for nm, a in zip(("tr", "rot", "tor", "sc_tor"), out):
            ref = t(z[f"score_{nm}_{step}"])
            assert (a - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-7, (nm, step)
    noise = sampler.draw_noise(scfg.actual_steps, G, int(d.tor_edge_mask.sum()), int(d.sc_torsion_edge_mask.sum()),
                               seed=int(z["noise_seed"]))
    assert torch.equal(noise.tr, t(z["noise_tr"])) and torch.equal(noise.sc, t(z["noise_sc"]))
    lig, a14 = sampler.sample(params, mcfg, scfg, copy.deepcopy(d), noise, t(T["atom14_to_group"]).long(), visualize=True)
    assert (lig - t(z["traj_lig"])).norm(dim=-1).max() < 1e-3
    assert (a14 - t(z["traj_atom14"])).norm(dim=-1).max() < 1e-3


@pytest.mark.parametrize("tag,over", [("ode", dict(type="ode")), ("no_random", dict(no_random=True))])
def test_sampler_modes_fixture(tag, over):
    """tests/golden/sampler_modes.npz: the reference's own sample() with `type='ode'` (scFlex.py:162-165,199-200) and with
    `no_random=True` (:167-183) on the batch of sampler.npz; the oracle follows both (0.0 at generation time)."""
    d, z = load_golden_batch()
    zm = np.load(os.path.join(GOLDEN, "sampler_modes.npz"))
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=int(zm["params_seed"]))
    scfg = schedule.default_sample_cfg(**over)
    G = d.num_graphs
    # a tape of ones: neither mode may read it
    noise = sampler.draw_noise(scfg.actual_steps, G, int(d.tor_edge_mask.sum()), int(d.sc_torsion_edge_mask.sum()), seed=1)
    for k in ("tr", "rot", "tor", "sc"):
        getattr(noise, k).fill_(1.0)
    lig, a14 = sampler.sample(params, mcfg, scfg, copy.deepcopy(d), noise, t(T["atom14_to_group"]).long(), visualize=True)
    assert (lig - t(zm[f"{tag}_traj_lig"])).norm(dim=-1).max() < 1e-3
    assert (a14 - t(zm[f"{tag}_traj_atom14"])).norm(dim=-1).max() < 1e-3


def test_time_schedule_guard_matches_the_reference():
    zm = np.load(os.path.join(GOLDEN, "sampler_modes.npz"))
    with pytest.raises(NotImplementedError) as e:
        schedule.t_schedule(schedule.default_sample_cfg(time_schedule="cosine"))
    assert str(e.value) == str(zm["time_schedule_error"])
