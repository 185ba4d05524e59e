"""Row f4: the MDN pose scorer's network forward -- KarmaDock.forward of
DiffBindFR/scoring/architecture/KarmaDock_sc.py (graph-transformer ligand encoder, GVP pocket encoder, mixture-density
head).  tests/golden/mdn.npz holds featurised synthetic inputs and what the REFERENCE's own architecture files computed
for them with seeded weights (tests/golden/make_golden.py::golden_mdn).  CPU: the oracle against the fixture.  GPU: the
HIP path (dbfr_mdn_*) against the fixture and against the oracle on ragged batches."""
import os

import numpy as np
import pytest
import torch

from oracle import mdn_scorer as oms
from tests.helpers import GOLDEN, rel_err


def fixture():
    z = np.load(os.path.join(GOLDEN, "mdn.npz"))
    d = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith(("ref_", "params_"))}
    return d, z


def test_oracle_matches_reference_fixture():
    d, z = fixture()
    P = oms.init_params(seed=int(z["params_seed"]))
    score, lig_s, pro_s = oms.forward(P, d)
    assert (lig_s - torch.from_numpy(z["ref_lig_s"])).abs().max() < 2e-5
    assert (pro_s - torch.from_numpy(z["ref_pro_s"])).abs().max() < 2e-5
    assert rel_err(score, torch.from_numpy(z["ref_score"])) < 1e-5


def test_pair_distance_quirks():
    """MDN_Block.compute_euclidean_distances_matrix: float64 expansion |x|^2 + |y|^2 - 2xy, NaN -> 10000, min over the 14
    slots with unused slots (zeros) NOT masked."""
    lig = torch.tensor([[1.0, 2.0, 3.0], [50.0, 0.0, 0.0]])
    res = torch.zeros(2, 14, 3)
    res[0, 0] = torch.tensor([1.0, 2.0, 3.0])          # coincident with ligand atom 0: distance 0 (or 10000 if d^2 < 0)
    res[0, 1:] = 100.0
    res[1, :2] = torch.tensor([[50.0, 3.0, 4.0], [60.0, 0.0, 0.0]])   # slots 2.. stay at the origin
    d = oms.pair_distance(lig, res)
    assert d.dtype == torch.float64 and d.shape == (2, 2)
    assert d[0, 0] in (0.0, 10000.0) and abs(d[1, 1] - 5.0) < 1e-9
    assert abs(d[0, 1] - (14.0 ** 0.5)) < 1e-9           # the origin slots of residue 1 are nearer than its real atoms


def test_product_parameter_names_match_the_oracle():
    from diffbindfr_amd import mdn
    assert mdn.param_shapes() == {k: tuple(v) for k, v in oms.param_shapes().items()}
