"""Property / known-answer tests of the e3nn restatement (parity UNPINNED boundary: no reference
vectors exist, see oracle/e3nn_lite.py) and of the library's own SO(3) tables against it."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import e3nn_lite as o3

torch.manual_seed(0)


def rand_rot(g):
    q = torch.randn(4, generator=g, dtype=torch.float64)
    q = q / q.norm()
    w, x, y, z = q
    return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)]),
                        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)]),
                        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)])])


def wigner_D(l, R):
    """D^l(R) from the harmonics themselves: Y(R v) = D Y(v) (least squares over random v)."""
    g = torch.Generator().manual_seed(l)
    v = torch.randn(64, 3, generator=g, dtype=torch.float64)
    A = o3.spherical_harmonics(l, v, True, "component")
    B = o3.spherical_harmonics(l, v @ R.T, True, "component")
    return torch.linalg.lstsq(A, B).solution.T


def test_known_answer_w3j():
    # SURVEY.md Appendix A.3b values
    assert abs(o3.wigner_3j(1, 1, 1)[0, 1, 2] - 1 / math.sqrt(6)) < 1e-12
    assert abs(o3.wigner_3j(1, 1, 1)[0, 2, 1] + 1 / math.sqrt(6)) < 1e-12
    for l in (1, 2):
        d = 1 / math.sqrt(2 * l + 1)
        assert torch.allclose(o3.wigner_3j(l, 0, l)[:, 0, :], d * torch.eye(2 * l + 1, dtype=torch.float64))
        assert torch.allclose(o3.wigner_3j(0, l, l)[0], d * torch.eye(2 * l + 1, dtype=torch.float64))
        assert torch.allclose(o3.wigner_3j(l, l, 0)[:, :, 0], d * torch.eye(2 * l + 1, dtype=torch.float64))
    c = o3.wigner_3j(1, 2, 1)
    assert abs(c[0, 2, 0] + 0.182574) < 1e-6 and abs(c[0, 4, 0] + 0.316228) < 1e-6 and abs(c[1, 1, 0] - 0.316228) < 1e-6
    assert abs(o3.wigner_3j(2, 2, 1)[0, 1, 0] + 0.182574) < 1e-6 and abs(o3.wigner_3j(2, 2, 1)[0, 3, 2] - 0.182574) < 1e-6
    assert abs(o3.wigner_3j(2, 2, 2)[0, 0, 2] + 0.239046) < 1e-6 and abs(o3.wigner_3j(2, 2, 2)[0, 1, 3] - 0.207020) < 1e-6


@pytest.mark.parametrize("ls", [(0, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1), (1, 2, 1), (2, 2, 0), (2, 2, 1),
                                (2, 1, 1), (1, 1, 2), (2, 2, 2)])
def test_w3j_rotation_invariant(ls):
    g = torch.Generator().manual_seed(3)
    R = rand_rot(g)
    C3 = o3.wigner_3j(*ls)
    D = [wigner_D(l, R) for l in ls]
    rot = torch.einsum("ijk,ia,jb,kc->abc", C3, D[0], D[1], D[2])
    assert (rot - C3).abs().max() < 1e-9
    assert abs(C3.norm() - 1) < 1e-12


def test_sh_normalisation_and_zero():
    v = torch.randn(10, 3, dtype=torch.float64)
    sh = o3.spherical_harmonics(o3.Irreps.spherical_harmonics(2), v, True, "component")
    assert torch.allclose(sh[:, 0], torch.ones(10, dtype=torch.float64))
    assert torch.allclose((sh[:, 1:4] ** 2).sum(-1), torch.full((10,), 3.0, dtype=torch.float64))
    assert torch.allclose((sh[:, 4:9] ** 2).sum(-1), torch.full((10,), 5.0, dtype=torch.float64))
    z = o3.spherical_harmonics(o3.Irreps.spherical_harmonics(2), torch.zeros(1, 3), True, "component")
    assert torch.all(z[:, 1:] == 0)          # zero vector stays zero (F.normalize)


def _irreps_D(irreps, R):
    blocks = []
    for mi in o3.Irreps(irreps):
        det = torch.linalg.det(R)
        D = wigner_D(mi.ir.l, R * det) * (det ** mi.ir.l if False else 1.0)
        par = (mi.ir.p if det < 0 else 1.0)
        for _ in range(mi.mul):
            blocks.append(D * par)
    return torch.block_diag(*blocks)


@pytest.mark.parametrize("improper", [False, True])
def test_fctp_equivariance(improper):
    g = torch.Generator().manual_seed(7)
    R = rand_rot(g) * (-1.0 if improper else 1.0)
    i, o = "4x0e + 4x1o + 4x1e + 4x0o", "4x0e + 4x1o + 4x1e + 4x0o"
    sh = o3.Irreps.spherical_harmonics(2)
    tp = o3.FullyConnectedTensorProduct(i, sh, o)
    z = 5
    x = torch.randn(z, o3.Irreps(i).dim, generator=g, dtype=torch.float64)
    v = torch.randn(z, 3, generator=g, dtype=torch.float64)
    w = torch.randn(z, tp.weight_numel, generator=g, dtype=torch.float64)
    Di, Do = _irreps_D(i, R), _irreps_D(o, R)
    y = tp(x, o3.spherical_harmonics(sh, v, True, "component"), w)
    y_rot = tp(x @ Di.T, o3.spherical_harmonics(sh, v @ R.T, True, "component"), w)
    assert (y_rot - y @ Do.T).abs().max() < 1e-9


def test_full_tensor_product_layout_and_equivariance():
    sh = o3.Irreps.spherical_harmonics(2)
    ft = o3.FullTensorProduct(sh, "2e")
    assert str(ft.irreps_out).startswith("1x0e+1x1o+1x1e") and ft.irreps_out.dim == 45
    g = torch.Generator().manual_seed(9)
    R = rand_rot(g)
    a, b = torch.randn(6, 3, generator=g, dtype=torch.float64), torch.randn(6, 3, generator=g, dtype=torch.float64)
    f = lambda a, b: ft(o3.spherical_harmonics(sh, a, True, "component"), o3.spherical_harmonics("2e", b, True, "component"))
    y, yr = f(a, b), f(a @ R.T, b @ R.T)
    low = o3.Irreps([mi for mi in ft.irreps_out if mi.ir.l <= 2])      # 0e 1o 1e 2o 2e 2e = first 22 components
    D = _irreps_D(low, R)
    assert (yr[:, :low.dim] - y[:, :low.dim] @ D.T).abs().max() < 1e-9


def test_weight_numel_table():
    # W per layer and head
    from oracle import score_model as sm
    cfg = sm.default_cfg()
    specs = sm.conv_specs(cfg)
    got = {k: o3.FullyConnectedTensorProduct(i, s, o).weight_numel for k, (i, s, o, _) in specs.items()}
    assert [got[f"lig_conv_layers.{l}"] for l in range(6)] == [2880, 3888, 4896, 7776, 7776, 7776]
    assert got["final_conv"] == 336 and got["tor_bond_conv"] == 6912 and got["sc_tor_bond_conv"] == 6912


# ---- the product's own SO(3) tables (libdbfr, host code: no GPU needed) against the oracle's
def test_library_wigner3j_matches_oracle():
    from diffbindfr_amd import lib as L
    lib = L.load()
    for ls in [(0, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1), (1, 2, 1), (2, 2, 0), (2, 2, 1), (2, 2, 2)]:
        n = (2 * ls[0] + 1) * (2 * ls[1] + 1) * (2 * ls[2] + 1)
        buf = (C.c_double * n)()
        assert lib.dbfr_wigner3j(*ls, buf) == 0
        got = torch.tensor(list(buf), dtype=torch.float64).reshape(2 * ls[0] + 1, 2 * ls[1] + 1, 2 * ls[2] + 1)
        assert (got - o3.wigner_3j(*ls)).abs().max() < 1e-12, ls


def test_library_path_tables_match_oracle():
    from diffbindfr_amd.score_model import conv_paths
    from oracle import score_model as sm
    cfg = sm.default_cfg()
    specs = sm.conv_specs(cfg)
    names = {0: "lig_conv_layers.0", 1: "lig_conv_layers.1", 2: "lig_conv_layers.2", 3: "lig_conv_layers.3",
             4: "final_conv", 5: "tor_bond_conv"}
    for kind, name in names.items():
        i, s, o, _ = specs[name]
        tp = o3.FullyConnectedTensorProduct(i, s, o)
        wn, tab = conv_paths(kind)
        assert wn == tp.weight_numel
        assert len(tab) == len(tp.instructions)
        for row, ins in zip(tab, tp.instructions):
            coeff = np.array([row[9]], dtype=np.int32).view(np.float32)[0]
            assert row[0] == ins.i1 and row[2] == ins.io and row[8] == ins.w_off, (name, row)
            assert abs(coeff - ins.coeff) < 1e-6
            # sh slot: kinds 0-4 share the oracle's numbering; tor convs keep only the 0e,1o,1e slots
            assert row[1] == ins.i2
