"""Developer script (GPU box): seeded fuzz of the HIP score network against the CPU oracle on random ragged batches.
    python tests/tools/fuzz_scores.py [n_cases]
Test infrastructure (imports oracle/)."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import diffbindfr_amd as dba
from diffbindfr_amd import synthetic
from oracle import sampler as osampler, schedule as osched, score_model as sm
from tests.helpers import namespace_to, rel_err
dev = torch.device("cuda:0")
mcfg = sm.default_cfg(); params = sm.init_params(mcfg, seed=1)
model = dba.TensorProductModelHIP({}).to(dev); model.load_state_dict(params, strict=True)
worst = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    rng = np.random.default_rng(1000 + case)
    items = []
    for g in range(int(rng.integers(1, 5))):
        na, nl = int(rng.integers(12, 160)), int(rng.integers(2, 40))
        pk, lg = synthetic.make_pocket(rng, na), synthetic.make_ligand(rng, nl)
        if rng.random() < 0.2:
            lg["tor_edge_mask"][:] = False; lg["rot_node_mask"] = lg["rot_node_mask"][:0]
        items.append((pk, lg) + synthetic.init_pose(rng, pk, lg, tr_sigma=float(rng.choice([0.5, 3.0, 10.0]))))
    d = synthetic.collate(items)
    step = int(rng.integers(0, 20))
    sc = osched.step_scalars(osched.default_sample_cfg(), step)
    dd = osampler.set_time(copy.deepcopy(d), sc, d.num_graphs)
    try:
        ref = sm.forward(params, mcfg, copy.deepcopy(dd))
    except Exception as e:
        print("case", case, "oracle failed:", type(e).__name__, str(e)[:80]); continue
    out = model(namespace_to(dd, dev))
    errs = [rel_err(a, b) if b.numel() else 0.0 for a, b in zip(out, ref)]
    worst = max(worst, max(errs))
    flag = "" if max(errs) < 1e-4 else "  <-- FAIL"
    print("case", case, "graphs", d.num_graphs, "NL", d.lig_pos.shape[0], "NA", d.rec_atm_pos.shape[0], "step", step, ["%.1e" % e for e in errs], flag)
print("worst", worst)
