"""CPU restatement of the 20-step reverse-SDE sampler ``DiffBindFR.sample``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
druglib/models/Docking/scFlex.py:104-250 step for step (deepcopy per step,
set_time, score network, Euler-Maruyama perturbations, ligand update, chi
update + side-chain rebuild), with two explicit inputs replacing the reference's
hidden global state so that runs are reproducible and comparable:

* ``noise``: the pre-drawn N(0,1) tape ``z_tr[T,G,3], z_rot[T,G,3], z_tor[T,n_tor],
  z_sc[T,n_sc]`` (the reference draws these from torch's global CPU generator in
  exactly this order each step, scFlex.py:167-183,202-205; nothing on the last step);
* ``torus_seed``: seed of the torus score-norm Monte-Carlo (unseeded in the reference).
"""
import copy
from types import SimpleNamespace

import numpy as np
import torch

from . import geometry, schedule, score_model


def draw_noise(actual_steps, G, n_tor, n_sc, seed, no_final_step_noise=True, none=False):
    """Tape drawn in the reference's order (tr, rot, tor, sc_tor per step).  ``none``: the all-zero tape of `type='ode'` /
    `no_random=True` runs (the reference draws nothing then, scFlex.py:162-183)."""
    g = torch.Generator().manual_seed(seed)
    z = SimpleNamespace(tr=torch.zeros(actual_steps, G, 3), rot=torch.zeros(actual_steps, G, 3),
                        tor=torch.zeros(actual_steps, n_tor), sc=torch.zeros(actual_steps, n_sc))
    for s in range(actual_steps):
        if none or (no_final_step_noise and s == actual_steps - 1):
            continue
        z.tr[s] = torch.normal(0, 1, size=(G, 3), generator=g)
        z.rot[s] = torch.normal(0, 1, size=(G, 3), generator=g)
        z.tor[s] = torch.normal(0, 1, size=(n_tor,), generator=g)
        z.sc[s] = torch.normal(0, 1, size=(n_sc,), generator=g)
    return z


def set_time(data, sc, G):
    """scFlex.py:104-122 given the per-step scalars ``sc`` (schedule.step_scalars)."""
    data.t = torch.tensor([float(sc.t)] * G, dtype=torch.float32)
    data.tr_sigma = torch.tensor([float(sc.tr_sigma)] * G, dtype=torch.float32)
    data.rot_score_norm = sc.rot_score_norm.repeat(G, 1)
    n_tor = int(data.tor_edge_mask.sum())
    data.tor_score_norm2 = sc.tor_score_norm2.repeat(n_tor)
    m = data.sc_torsion_edge_mask
    data.sc_tor_score_norm2 = sc.tor_score_norm2.view(1, 1).expand(m.shape).float() * m
    return data


def sde_step(data, sc, scores, z, t_idx, atom14_to_group):
    """scFlex.py:154-230: perturbations + the two geometry updates, in place on ``data``."""
    tr_score, rot_score, tor_score, sc_tor_score = scores
    dt = sc.dt
    ode = getattr(sc, "ode", False)
    if ode:                                           # scFlex.py:162-165: probability-flow ODE, no noise
        tr_perturb = 0.5 * sc.tr_g ** 2 * tr_score * dt
        rot_perturb = 0.5 * sc.rot_g ** 2 * rot_score * dt
        tor_perturb = 0.5 * sc.tor_g ** 2 * tor_score * dt
    else:                                             # :166-183.  `no_random`: z = 0 whatever the tape holds; the noise-free LAST step
        zero = getattr(sc, "no_random", False)        # (no_final_step_noise) is the tape's business: draw_noise leaves zeros there
        tr_perturb = sc.tr_g ** 2 * tr_score * dt + sc.tr_g * np.sqrt(dt) * (torch.zeros_like(z.tr[t_idx]) if zero else z.tr[t_idx])
        rot_perturb = sc.rot_g ** 2 * rot_score * dt + sc.rot_g * np.sqrt(dt) * (torch.zeros_like(z.rot[t_idx]) if zero else z.rot[t_idx])
        tor_perturb = sc.tor_g ** 2 * tor_score * dt + sc.tor_g * np.sqrt(dt) * (torch.zeros_like(z.tor[t_idx]) if zero else z.tor[t_idx])
    data.lig_pos = geometry.update_batchlig_pos(
        tr_perturb, rot_perturb, tor_perturb, data.lig_pos, data.lig_edge_index,
        data.tor_edge_mask, data.rot_node_mask, batch=data.lig_node_batch)
    if ode:                                           # :199-200
        sc_perturb = 0.5 * sc.sc_tor_g ** 2 * sc_tor_score * dt
    else:
        sc_perturb = sc.sc_tor_g ** 2 * sc_tor_score * dt + sc.sc_tor_g * np.sqrt(dt) * (torch.zeros_like(z.sc[t_idx]) if getattr(sc, "no_random", False) else z.sc[t_idx])
    chi = data.torsion_angle[:, 1:]
    chi[data.sc_torsion_edge_mask] = chi[data.sc_torsion_edge_mask] + sc_perturb
    data.torsion_angle[:, 1:] = chi
    atom14 = geometry.build_atom14(data.sequence, data.backbone_transl, data.backbone_rots,
                                   data.default_frame, data.rigid_group_positions,
                                   data.torsion_angle, atom14_to_group)
    mask = data.atom14_mask.bool()
    atom14 = atom14 * mask.unsqueeze(-1)
    data.rec_atm_pos = atom14[mask]
    return atom14, SimpleNamespace(tr=tr_perturb, rot=rot_perturb, tor=tor_perturb, sc=sc_perturb)


def sample(params, mcfg, scfg, data, noise, atom14_to_group, torus_seed=0, visualize=False,
           score_tap=None, first_step=0):
    """Returns (lig_traj [T,N_l_total,3], atom14_traj [T,N_res_total,14,3]) with T=1
    unless ``visualize`` (scFlex.py:193-194,227-230).  ``score_tap(t_idx, scores)`` is an
    optional callback (used by parity tests to compare per-step scores).  ``first_step``: resume the
    reverse SDE at that step from the state ``data`` holds (tests/test_examples.py: the oracle re-run from
    the library's own coordinates after a hard-cutoff event); the noise tape is indexed by the absolute step."""
    G = int(data.lig_node_batch.max().item()) + 1
    lig_out, a14_out = [], []
    with torch.no_grad():
        for t_idx in range(first_step, scfg.actual_steps):
            sc = schedule.step_scalars(scfg, t_idx, torus_seed)
            _data = set_time(copy.deepcopy(data), sc, G)
            scores = score_model.forward(params, mcfg, _data)
            if score_tap is not None:
                score_tap(t_idx, scores)
            atom14, _ = sde_step(data, sc, scores, noise, t_idx, atom14_to_group)
            if visualize or t_idx == scfg.actual_steps - 1:
                lig_out.append(data.lig_pos.clone())
                a14_out.append(atom14.clone())
    return torch.stack(lig_out, 0), torch.stack(a14_out, 0)
