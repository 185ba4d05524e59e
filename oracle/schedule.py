"""Reverse-SDE schedule and score-norm tables (CPU restatement).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
* druglib/models/Docking/scFlex.py:83-122,154-161,197-198 (t schedule, sigma_fn, g),
* druglib/utils/geometry_utils/so3.py:29-62,93-101,144-149 (truncated series ->
  ``_exp_score_norms``; deterministic),
* druglib/utils/geometry_utils/torus.py:25-31,36-66,73-83,102-115
  (``score_norm_`` = 10 000-sample Monte-Carlo of the tabulated score, drawn
  from numpy's *unseeded* global RNG in the reference => the oracle takes an
  explicit seed and the resulting table travels as a tape into both the oracle
  sampler and the HIP sampler).
Only the rows of the big tables that a query touches are evaluated (the
reference builds the full 5001x5001 torus table, several GB transiently).
"""
import math
from types import SimpleNamespace

import numpy as np
import torch


def default_sample_cfg(**over):
    """DiffBindFR/configs/diffbindfr_ts.py:2-10,144-162."""
    cfg = dict(type="sde", time_schedule="linear", inference_steps=22, actual_steps=20, eps=1e-5,
               no_final_step_noise=True, no_random=False,
               tr_sigma_min=0.1, tr_sigma_max=6, rot_sigma_min=0.03, rot_sigma_max=1.55,
               tor_sigma_min=0.0314, tor_sigma_max=3.14, sc_tor_sigma_min=0.0314, sc_tor_sigma_max=3.14)
    cfg.update(over)
    return SimpleNamespace(**cfg)


def t_schedule(cfg):
    """scFlex.py:83-91."""
    if cfg.time_schedule != "linear":
        raise NotImplementedError("Current time schedule only supports `linear`.")
    return torch.linspace(1, cfg.eps, cfg.inference_steps + 1)


def sigma_fn(cfg, t):
    """scFlex.py:93-102 with t a 0-d fp32 tensor (as in the reference's loop)."""
    return (cfg.tr_sigma_min ** (1 - t) * cfg.tr_sigma_max ** t,
            cfg.rot_sigma_min ** (1 - t) * cfg.rot_sigma_max ** t,
            cfg.tor_sigma_min ** (1 - t) * cfg.tor_sigma_max ** t,
            cfg.sc_tor_sigma_min ** (1 - t) * cfg.sc_tor_sigma_max ** t)


# --------------------------------------------------------------------------- so3
SO3_MIN_EPS, SO3_MAX_EPS, SO3_N_EPS, SO3_X_N = 0.01, 2, 1000, 2000


def so3_eps_index(eps):
    eps = np.asarray(eps)
    idx = (np.log10(eps) - np.log10(SO3_MIN_EPS)) / (np.log10(SO3_MAX_EPS) - np.log10(SO3_MIN_EPS)) * SO3_N_EPS
    return np.clip(np.around(idx).astype(int), a_min=0, a_max=SO3_N_EPS - 1)


_SO3_CACHE = {}


def so3_exp_score_norm(eps_idx, L=2000):
    """One entry of so3._exp_score_norms (so3.py:93-101)."""
    eps_idx = int(eps_idx)
    if eps_idx in _SO3_CACHE:
        return _SO3_CACHE[eps_idx]
    eps = (10 ** np.linspace(np.log10(SO3_MIN_EPS), np.log10(SO3_MAX_EPS), SO3_N_EPS))[eps_idx]
    omega = np.linspace(0, np.pi, SO3_X_N + 1)[1:]
    p = 0
    for l in range(L):  # so3._expansion
        p += (2 * l + 1) * np.exp(-l * (l + 1) * eps ** 2) * np.sin(omega * (l + 1 / 2)) / np.sin(omega / 2)
    pdf = p * (1 - np.cos(omega)) / np.pi          # so3._density(marginal=True)
    d = 0
    for l in range(L):  # so3._score
        hi = np.sin(omega * (l + 1 / 2))
        dhi = (l + 1 / 2) * np.cos(omega * (l + 1 / 2))
        lo = np.sin(omega / 2)
        dlo = 1 / 2 * np.cos(omega / 2)
        d += (2 * l + 1) * np.exp(-l * (l + 1) * eps ** 2) * (lo * dhi - hi * dlo) / lo ** 2
    score = d / p
    val = np.sqrt(np.sum(score ** 2 * pdf) / np.sum(pdf) / np.pi)
    _SO3_CACHE[eps_idx] = val
    return val


def so3_score_norm(eps):
    """so3.score_norm (so3.py:144-149): float32 tensor, one per eps."""
    idx = so3_eps_index(eps)
    return torch.from_numpy(np.asarray([so3_exp_score_norm(i) for i in np.atleast_1d(idx)])).float()


# --------------------------------------------------------------------------- torus
TOR_X_MIN, TOR_X_N = 1e-5, 5000
TOR_SIGMA_MIN, TOR_SIGMA_MAX, TOR_SIGMA_N = 3e-3, 2, 5000


def torus_sigma_index(sigma):
    s = np.log(np.asarray(sigma) / np.pi)          # dtype of the caller kept (fp32 in scFlex.py:116-120)
    s = (s - np.log(TOR_SIGMA_MIN)) / (np.log(TOR_SIGMA_MAX) - np.log(TOR_SIGMA_MIN)) * TOR_SIGMA_N
    return np.round(np.clip(s, 0, TOR_SIGMA_N)).astype(int)


_TORUS_CACHE = {}


def torus_score_norm_entry(sigma_idx, seed=0, n_samples=10000, N=100):
    """One entry of torus.score_norm_ (torus.py:102-106), seeded.

    Row ``sigma_idx`` of the reference's p_/score_ tables (torus.py:25-66), then the
    Monte-Carlo mean of score^2 over ``n_samples`` wrapped-normal draws."""
    key = (int(sigma_idx), int(seed), n_samples)
    if key in _TORUS_CACHE:
        return _TORUS_CACHE[key]
    x = 10 ** np.linspace(np.log10(TOR_X_MIN), 0, TOR_X_N + 1) * np.pi
    sigma = (10 ** np.linspace(np.log10(TOR_SIGMA_MIN), np.log10(TOR_SIGMA_MAX), TOR_SIGMA_N + 1) * np.pi)[sigma_idx]
    p_ = 0
    g_ = 0
    for i in range(-N, N + 1):
        e = np.exp(-(x + 2 * np.pi * i) ** 2 / 2 / sigma ** 2)
        p_ += e
        g_ += (x + 2 * np.pi * i) / sigma ** 2 * e
    score_row = g_ / p_
    rng = np.random.default_rng([int(seed), int(sigma_idx)])
    s = sigma * rng.standard_normal(n_samples)
    s = (s + np.pi) % (2 * np.pi) - np.pi                      # torus.sample
    sign = np.sign(s)
    xs = np.log(np.abs(s) / np.pi)
    xs = (xs - np.log(TOR_X_MIN)) / (0 - np.log(TOR_X_MIN)) * TOR_X_N
    xs = np.round(np.clip(xs, 0, TOR_X_N)).astype(int)
    sc = -sign * score_row[xs]                                  # torus.score
    val = float((sc ** 2).mean())
    _TORUS_CACHE[key] = val
    return val


def torus_score_norm(sigma, seed=0):
    """torus.score_norm (torus.py:109-115): float64 ndarray shaped like sigma."""
    if torch.is_tensor(sigma):
        sigma = sigma.cpu().numpy()
    idx = torus_sigma_index(sigma)
    flat = np.asarray([torus_score_norm_entry(i, seed) for i in np.atleast_1d(idx).ravel()])
    return flat.reshape(np.shape(idx))


# --------------------------------------------------------------------------- per-step tape
def step_scalars(cfg, t_idx, torus_seed=0):
    """Everything scalar the reference derives per step (scFlex.py:104-122,146-161,197-198),
    computed with the same tensor / numpy mixture so the fp32 values agree bit for bit."""
    ts = t_schedule(cfg)
    t = ts[t_idx]
    dt = ts[t_idx] - ts[t_idx + 1]
    tr_s, rot_s, tor_s, sc_s = sigma_fn(cfg, t)
    tr_g = tr_s * np.sqrt(2 * np.log(cfg.tr_sigma_max / cfg.tr_sigma_min))
    rot_g = 2 * rot_s * np.sqrt(np.log(cfg.rot_sigma_max / cfg.rot_sigma_min))
    tor_g = tor_s * np.sqrt(2 * np.log(cfg.tor_sigma_max / cfg.tor_sigma_min))
    sc_g = sc_s * np.sqrt(2 * np.log(cfg.sc_tor_sigma_max / cfg.sc_tor_sigma_min))
    rot_norm = so3_score_norm(np.array([rot_s]))                       # [1] f32
    # scFlex.py:116 uses sc_tor_sigma for the *ligand* torsion norm too (quirk kept)
    tor_norm2 = torch.from_numpy(torus_score_norm(torch.ones(1) * sc_s, torus_seed)).float()
    last = cfg.no_final_step_noise and t_idx == cfg.actual_steps - 1
    ode = cfg.type == "ode"                                            # scFlex.py:162: anything else takes the SDE branch
    return SimpleNamespace(t=t, dt=dt, tr_sigma=tr_s, rot_sigma=rot_s, tor_sigma=tor_s, sc_tor_sigma=sc_s,
                           tr_g=tr_g, rot_g=rot_g, tor_g=tor_g, sc_tor_g=sc_g, rot_score_norm=rot_norm,
                           tor_score_norm2=tor_norm2, ode=ode, no_random=bool(cfg.no_random),
                           noise_free=bool(ode or cfg.no_random or last))
