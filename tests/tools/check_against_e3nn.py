"""Run on any machine that HAS e3nn==0.5.1 installed (the build container has not): pins the one boundary
this repo cannot pin offline -- oracle/e3nn_lite.py against the real e3nn arithmetic the reference calls.

    pip install e3nn==0.5.1 && python tests/tools/check_against_e3nn.py

Checks: spherical harmonics (component normalisation), every Wigner-3j tensor the model touches (incl. the
global sign), FullyConnectedTensorProduct (weight layout, path normalisation) for the 6 conv signatures, and
FullTensorProduct(sh, "2e").  Exit status 0 = e3nn_lite reproduces e3nn to 1e-6.

It also FREEZES what the real e3nn computed (inputs + outputs) as tests/golden/e3nn_051.npz; commit that file and
tests/test_e3nn_lite.py::test_against_frozen_e3nn_outputs_when_available pins the restatement from then on, on machines
without e3nn too.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import e3nn_lite as lite  # noqa: E402

try:
    from e3nn import o3
except ImportError:
    print("e3nn is not installed: nothing to check against (parity at this boundary stays unpinned)")
    sys.exit(2)

import numpy as np  # noqa: E402

torch.manual_seed(0)
worst = 0.0
frozen = {}


def cmp(name, a, b):
    global worst
    err = (a.double() - b.double()).abs().max().item()
    worst = max(worst, err)
    print(f"{name:60s} max|d| = {err:.2e}")


v = torch.randn(50, 3)
sh = o3.Irreps.spherical_harmonics(2)
frozen["sh_in"] = v.numpy()
frozen["sh_out"] = o3.spherical_harmonics(sh, v, normalize=True, normalization="component").numpy()
cmp("spherical_harmonics l<=2 component", lite.spherical_harmonics(lite.Irreps.spherical_harmonics(2), v, True, "component"),
    o3.spherical_harmonics(sh, v, normalize=True, normalization="component"))
for ls in [(0, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1), (1, 1, 2), (1, 2, 1), (2, 1, 1), (2, 2, 0), (2, 2, 1), (2, 2, 2)]:
    frozen["w3j_%d%d%d" % ls] = o3.wigner_3j(*ls).numpy()
    cmp(f"wigner_3j{ls}", lite.wigner_3j(*ls), o3.wigner_3j(*ls).double())
full = "48x0e + 12x1o + 12x1e + 48x0o"
sigs = [("48x0e", sh, "48x0e + 12x1o"), ("48x0e + 12x1o", sh, "48x0e + 12x1o + 12x1e"),
        ("48x0e + 12x1o + 12x1e", sh, full), (full, sh, full), (full, sh, "2x1o + 2x1e")]
for i, s, o in sigs:
    a, b = lite.FullyConnectedTensorProduct(i, str(s), o), o3.FullyConnectedTensorProduct(i, s, o, shared_weights=False)
    assert a.weight_numel == b.weight_numel, (i, o)
    x = torch.randn(7, o3.Irreps(i).dim)
    y = o3.spherical_harmonics(s, torch.randn(7, 3), normalize=True, normalization="component")
    w = torch.randn(7, b.weight_numel)
    k = f"fctp{len([q for q in frozen if q.endswith('_x')])}"
    with torch.no_grad():
        frozen.update({k + "_x": x.numpy(), k + "_y": y.numpy(), k + "_w": w.numpy(), k + "_out": b(x, y, w).numpy(),
                       k + "_irreps": np.asarray([i, str(s), o])})
    cmp(f"FCTP {i} -> {o}", a(x, y, w), b(x, y, w))
ft_l, ft_e = lite.FullTensorProduct(str(sh), "2e"), o3.FullTensorProduct(sh, "2e")
assert str(ft_e.irreps_out).replace(" ", "") == str(ft_l.irreps_out), (ft_e.irreps_out, ft_l.irreps_out)
e = o3.spherical_harmonics(sh, torch.randn(9, 3), normalize=True, normalization="component")
b2 = o3.spherical_harmonics("2e", torch.randn(9, 3), normalize=True, normalization="component")
frozen.update(ft_a=e.numpy(), ft_b=b2.numpy(), ft_out=ft_e(e, b2).detach().numpy())
cmp("FullTensorProduct(sh, 2e)", ft_l(e, b2), ft_e(e, b2))
a, b = lite.FullyConnectedTensorProduct(full, str(ft_l.irreps_out), "48x0o + 48x0e"), \
    o3.FullyConnectedTensorProduct(full, ft_e.irreps_out, "48x0o + 48x0e", shared_weights=False)
x, w = torch.randn(5, 168), torch.randn(5, b.weight_numel)
cmp("FCTP tor_bond_conv", a(x, ft_l(e[:5], b2[:5]), w), b(x, ft_e(e[:5], b2[:5]), w))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "e3nn_051.npz")
np.savez_compressed(out, e3nn_version=np.asarray(getattr(__import__("e3nn"), "__version__", "?")), **frozen)
print("frozen real-e3nn outputs ->", os.path.normpath(out))
print("worst", worst)
sys.exit(0 if worst < 1e-6 else 1)
