import sys, os, time, copy, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import diffbindfr_amd as dba
from diffbindfr_amd import lib as L, synthetic
from oracle import score_model as sm, sampler as osampler, schedule as osched
from helpers import *
dev = torch.device("cuda:0")
mcfg = sm.default_cfg(); params = sm.init_params(mcfg, seed=0)
model = dba.TensorProductModelHIP({}).to(dev); model.load_state_dict(params, strict=True)
d = synthetic.make_batch(2, n_complex=2, poses=3, seed=31)
sc = osched.step_scalars(osched.default_sample_cfg(), 4)
dd = osampler.set_time(copy.deepcopy(d), sc, d.num_graphs)
ref = sm.forward(params, mcfg, copy.deepcopy(dd))
def rel(a, b): return float((a.cpu() - b).abs().max() / b.abs().max())
for mode in ("f32", "split_f16", "reduce_first"):
    model.set_gemm(mode)
    out = model(namespace_to(copy.deepcopy(dd), dev))
    print(mode, model.gemm_mode(), [f"{rel(o, r):.2e}" for o, r in zip(out, ref) if o is not None], flush=True)
