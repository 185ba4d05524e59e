"""Host-side reverse-SDE schedule: per-step scalars handed to dbfr_sample.

Mirrors the reference's host arithmetic so the fp32 scalars entering the device
are the ones the reference would use:
  t / dt / sigma / g          druglib/models/Docking/scFlex.py:83-102,146-161,197-198
  so3 score norm              druglib/utils/geometry_utils/so3.py:29-62,93-101,144-149
  torus score norm            druglib/utils/geometry_utils/torus.py:25-66,73-115
The so3 entry is a deterministic truncated series; the torus entry is the
reference's Monte-Carlo estimator (10 000 wrapped-normal draws of the tabulated
score), which the reference draws from numpy's unseeded global RNG at import --
here it takes an explicit seed so that runs are reproducible.  Only the table rows
a query touches are evaluated (the reference materialises 1000x2000 and 5001x5001
tables at import).
"""
import ctypes
from functools import lru_cache
from types import SimpleNamespace

import numpy as np
import torch

from . import lib as L

SAMPLE_DEFAULTS = dict(  # DiffBindFR/configs/diffbindfr_ts.py:2-10,144-162
    type="sde", time_schedule="linear", inference_steps=22, actual_steps=20, eps=1e-5,
    no_final_step_noise=True, no_random=False, tr_sigma_min=0.1, tr_sigma_max=6, rot_sigma_min=0.03,
    rot_sigma_max=1.55, tor_sigma_min=0.0314, tor_sigma_max=3.14, sc_tor_sigma_min=0.0314, sc_tor_sigma_max=3.14)


def sample_cfg(cfg=None, **over):
    d = dict(SAMPLE_DEFAULTS)
    if cfg is not None:
        d.update({k: (cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)) for k in SAMPLE_DEFAULTS
                  if (k in cfg if isinstance(cfg, dict) else hasattr(cfg, k))})
    d.update(over)
    return SimpleNamespace(**d)


@lru_cache(maxsize=None)
def so3_exp_score_norm(eps_idx, L_terms=2000, n_eps=1000, x_n=2000, min_eps=0.01, max_eps=2.0):
    eps = (10 ** np.linspace(np.log10(min_eps), np.log10(max_eps), n_eps))[eps_idx]
    om = np.linspace(0, np.pi, x_n + 1)[1:]
    l = np.arange(L_terms)[:, None]
    w = (2 * l + 1) * np.exp(-l * (l + 1) * eps ** 2)
    hi, lo = np.sin(om * (l + 0.5)), np.sin(om / 2)
    # sequential accumulation over l like the reference's python loop (same rounding order)
    p = np.zeros_like(om)
    d = np.zeros_like(om)
    term_p = w * hi / lo
    term_d = w * (lo * ((l + 0.5) * np.cos(om * (l + 0.5))) - hi * (0.5 * np.cos(om / 2))) / lo ** 2
    for k in range(L_terms):
        p += term_p[k]
        d += term_d[k]
    pdf = p * (1 - np.cos(om)) / np.pi
    score = d / p
    return float(np.sqrt(np.sum(score ** 2 * pdf) / np.sum(pdf) / np.pi))


def so3_score_norm(rot_sigma):
    eps = np.asarray(rot_sigma)
    idx = (np.log10(eps) - np.log10(0.01)) / (np.log10(2) - np.log10(0.01)) * 1000
    idx = int(np.clip(np.around(idx).astype(int), 0, 999))
    return np.float32(so3_exp_score_norm(idx))


@lru_cache(maxsize=None)
def torus_score_norm_entry(sigma_idx, seed=0, n_samples=10000, n_wrap=100):
    x_min, x_n, s_min, s_max, s_n = 1e-5, 5000, 3e-3, 2, 5000
    x = 10 ** np.linspace(np.log10(x_min), 0, x_n + 1) * np.pi
    sigma = (10 ** np.linspace(np.log10(s_min), np.log10(s_max), s_n + 1) * np.pi)[sigma_idx]
    p_ = 0
    g_ = 0
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(-n_wrap, n_wrap + 1):
            e = np.exp(-(x + 2 * np.pi * i) ** 2 / 2 / sigma ** 2)
            p_ += e
            g_ += (x + 2 * np.pi * i) / sigma ** 2 * e
        row = g_ / p_
    rng = np.random.default_rng([int(seed), int(sigma_idx)])
    s = sigma * rng.standard_normal(n_samples)
    s = (s + np.pi) % (2 * np.pi) - np.pi
    sign = np.sign(s)
    xs = np.log(np.abs(s) / np.pi)
    xs = (xs - np.log(x_min)) / (0 - np.log(x_min)) * x_n
    xs = np.round(np.clip(xs, 0, x_n)).astype(int)
    return float(((-sign * row[xs]) ** 2).mean())


def torus_score_norm(sigma, seed=0):
    s = np.log(np.asarray(sigma) / np.pi)
    s = (s - np.log(3e-3)) / (np.log(2) - np.log(3e-3)) * 5000
    idx = int(np.round(np.clip(s, 0, 5000)).astype(int))
    return np.float32(torus_score_norm_entry(idx, seed))


def steps(cfg, torus_seed=0):
    """list of per-step scalar records + the ctypes dbfr_step array (actual_steps long).

    ``cfg.type == 'ode'`` (scFlex.py:162-165,199-200): the probability-flow step ``0.5 g^2 score dt`` without noise -- on the tape the
    drift factor becomes ``0.5 g^2`` (an exact halving: the device's ``g2 * score * dt`` then IS the reference's
    ``0.5 * g ** 2 * score * dt``, rounding for rounding), the noise factor 0, and every step is noise free (nothing is drawn, as in
    the reference).  Any other ``type`` is the SDE, like the reference's ``else``.  ``cfg.no_random`` (:167-183): SDE drift with z = 0.
    ``time_schedule`` other than 'linear' raises like scFlex.py:91."""
    if cfg.time_schedule != "linear":
        raise NotImplementedError("Current time schedule only supports `linear`.")
    assert cfg.actual_steps <= cfg.inference_steps, "actual steps should <= inference steps"      # scFlex.py:137
    ode = cfg.type == "ode"
    drift = 0.5 if ode else 1.0
    ts = torch.linspace(1, cfg.eps, cfg.inference_steps + 1)
    recs = []
    arr = (L.Step * cfg.actual_steps)()
    for i in range(cfg.actual_steps):
        t, dt = ts[i], ts[i] - ts[i + 1]
        tr_s = cfg.tr_sigma_min ** (1 - t) * cfg.tr_sigma_max ** t
        rot_s = cfg.rot_sigma_min ** (1 - t) * cfg.rot_sigma_max ** t
        tor_s = cfg.tor_sigma_min ** (1 - t) * cfg.tor_sigma_max ** t
        sc_s = cfg.sc_tor_sigma_min ** (1 - t) * cfg.sc_tor_sigma_max ** t
        tr_g = tr_s * np.sqrt(2 * np.log(cfg.tr_sigma_max / cfg.tr_sigma_min))
        rot_g = 2 * rot_s * np.sqrt(np.log(cfg.rot_sigma_max / cfg.rot_sigma_min))
        tor_g = tor_s * np.sqrt(2 * np.log(cfg.tor_sigma_max / cfg.tor_sigma_min))
        sc_g = sc_s * np.sqrt(2 * np.log(cfg.sc_tor_sigma_max / cfg.sc_tor_sigma_min))
        sq = np.sqrt(dt) * (0.0 if ode else 1.0)
        r = SimpleNamespace(
            t=float(t), dt=float(dt), tr_sigma=float(tr_s), rot_sigma=float(rot_s), tor_sigma=float(tor_s),
            sc_tor_sigma=float(sc_s), rot_score_norm=float(so3_score_norm(np.array([rot_s]))),
            # scFlex.py:116: the ligand torsion norm is looked up with sc_tor_sigma (quirk kept)
            tor_score_norm2=float(torus_score_norm((torch.ones(1) * sc_s).numpy(), torus_seed)),
            tr_g2=float(drift * tr_g ** 2), tr_gsdt=float(tr_g * sq), rot_g2=float(drift * rot_g ** 2), rot_gsdt=float(rot_g * sq),
            tor_g2=float(drift * tor_g ** 2), tor_gsdt=float(tor_g * sq), sc_g2=float(drift * sc_g ** 2), sc_gsdt=float(sc_g * sq),
            noise_free=bool(ode or cfg.no_random or (cfg.no_final_step_noise and i == cfg.actual_steps - 1)))
        recs.append(r)
        for k, _ in L.Step._fields_:
            setattr(arr[i], k, getattr(r, k))
    return recs, arr
