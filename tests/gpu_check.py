"""Developer script (GPU box): quick parity printout of the HIP path vs the oracle."""
import copy
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.helpers import load_golden_batch, namespace_to, rel_err  # noqa
import diffbindfr_amd as dba  # noqa
from diffbindfr_amd import synthetic  # noqa
from diffbindfr_amd.packing import PackedBatch  # noqa
from oracle import sampler as osampler, schedule as osched, score_model as sm  # noqa

dev = torch.device("cuda:0")
mcfg = sm.default_cfg()
params = sm.init_params(mcfg, seed=1)
model = dba.TensorProductModelHIP({}).to(dev)
model.load_state_dict(params, strict=True)
d, z = load_golden_batch()
G = d.num_graphs
scfg = osched.default_sample_cfg()
for step in (0, 10, 19):
    sc = osched.step_scalars(scfg, step)
    dd = osampler.set_time(copy.deepcopy(d), sc, G)
    ref = [torch.from_numpy(z[f"score_{n}_{step}"]) for n in ("tr", "rot", "tor", "sc_tor")]
    out = model(namespace_to(dd, dev))
    print(f"step {step}: rel err tr/rot/tor/sc =", [f"{rel_err(a, b):.2e}" for a, b in zip(out, ref)])
    if step == 0:
        print("  tr hip", out[0][0].cpu().numpy(), "ref", ref[0][0].numpy())

# sampler
samp = dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})
pb = PackedBatch(namespace_to(d, dev), dev)
noise = {k: torch.from_numpy(z[f"noise_{k}"]).to(dev).contiguous() for k in ("tr", "rot", "tor", "sc")}
t0 = time.time()
lig, a14 = samp.sample_packed(pb, noise, visualize=True)
torch.cuda.synchronize()
print("sample time", time.time() - t0)
ref_l, ref_a = torch.from_numpy(z["traj_lig"]), torch.from_numpy(z["traj_atom14"])
for s in (0, 1, 5, 10, 19):
    dl = (lig[s].cpu() - ref_l[s]).norm(dim=-1)
    da = (a14[s].cpu() - ref_a[s]).norm(dim=-1)
    print(f"traj step {s}: lig max dev {dl.max():.3e} A   atom14 max dev {da.max():.3e} A")
