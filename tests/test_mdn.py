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


# ------------------------------------------------------------------------------------------------ GPU
def _hip_model(dev, seed):
    from diffbindfr_amd import mdn
    P = oms.init_params(seed=seed)
    m = mdn.KarmaDockHIP().to(dev)
    m.load_state_dict(P, strict=True)
    return P, m


def _to(d, dev):
    return {k: v.to(dev) for k, v in d.items()}


@pytest.mark.gpu
def test_gpu_scorer_matches_reference_fixture():
    """Scores and both embeddings of the HIP forward against what the reference's KarmaDock computed (mdn.npz).
    Tolerance: embeddings 1e-4 absolute (fp32 sums in another order), scores 1e-4 relative."""
    dev = torch.device("cuda:0")
    d, z = fixture()
    P, m = _hip_model(dev, int(z["params_seed"]))
    score, lig_s, pro_s = m.score(_to(d, dev), return_embeddings=True)
    torch.cuda.synchronize()
    assert (lig_s.cpu() - torch.from_numpy(z["ref_lig_s"])).abs().max() < 1e-4
    assert (pro_s.cpu() - torch.from_numpy(z["ref_pro_s"])).abs().max() < 1e-4
    assert rel_err(score, torch.from_numpy(z["ref_score"])) < 1e-4, (score.cpu(), z["ref_score"])


@pytest.mark.gpu
def test_gpu_scorer_ragged_batches_vs_oracle_and_reuse_of_ligand_embeddings():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    dev = torch.device("cuda:0")
    P, m = _hip_model(dev, 7)
    rng = np.random.default_rng(5)
    from tests.test_mdn_inputs import mdn_inputs
    for sizes in ([(4, 31)], [(30, 140), (6, 33), (21, 75), (12, 90)], [(35, 105)] * 3):
        d = mdn_inputs(rng, sizes, coincident=False)
        ref, lig_ref, pro_ref = oms.forward(P, d)
        score, lig_s, pro_s = m.score(_to(d, dev), return_embeddings=True)
        assert (lig_s.cpu() - lig_ref).abs().max() < 1e-4 and (pro_s.cpu() - pro_ref).abs().max() < 1e-4
        assert rel_err(score, ref) < 1e-4, (sizes, score.cpu(), ref)
        # pose-independent ligand embeddings handed back in: same scores, bit for bit
        again = m.score(_to(d, dev), lig_s=lig_s)
        assert torch.equal(again, score)
        assert torch.equal(m.score(_to(d, dev)), score)            # reproducible
    # graph order does not matter
    d = mdn_inputs(rng, [(9, 40), (15, 52)], coincident=False)
    a = m.score(_to(d, dev)).cpu()
    sw = mdn_inputs(np.random.default_rng(99), [(15, 52)], coincident=False)   # unrelated batch in between
    m.score(_to(sw, dev))
    assert torch.equal(m.score(_to(d, dev)).cpu(), a)


@pytest.mark.gpu
def test_gpu_scorer_refuses_cpu_tensors_and_bad_state():
    from diffbindfr_amd import lib as L, mdn
    d, z = fixture()
    m = mdn.KarmaDockHIP()
    with pytest.raises(L.DbfrError):
        m.score(d)
