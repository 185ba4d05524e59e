"""Shared helpers of the parity tests (test infrastructure; may import oracle/)."""
import copy
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden_batch(path=None):
    """The small collated batch frozen in tests/golden/sampler.npz -> namespace of tensors."""
    z = np.load(path or os.path.join(GOLDEN, "sampler.npz"))
    d = SimpleNamespace()
    G = int(z["num_graphs"])
    for k in z.files:
        if k.startswith(("score_", "noise_", "traj_", "params_", "rot_node_mask_")) or k == "num_graphs":
            continue
        setattr(d, k, torch.from_numpy(z[k]))
    d.rot_node_mask = [torch.from_numpy(z[f"rot_node_mask_{g}"]) for g in range(G)]
    d.num_graphs = G
    return d, z


def namespace_to(d, device):
    o = SimpleNamespace()
    for k, v in vars(d).items():
        if torch.is_tensor(v):
            setattr(o, k, v.to(device))
        elif isinstance(v, list):
            setattr(o, k, [x.to(device) if torch.is_tensor(x) else x for x in v])
        else:
            setattr(o, k, v)
    return o


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
