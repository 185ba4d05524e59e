"""Developer: accuracy of DBFR_GEMM_REDUCE_FIRST against a float64 evaluation (segment sums per target and output column), next to the
fp32-instruction kernel (k_conv) and the per-edge split-fp16 kernel, over the weight distributions of tests/test_gpu_parity.py."""
import ctypes as C, os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import diffbindfr_amd as dba
from diffbindfr_amd import lib as L
from oracle import e3nn_lite as o3, score_model as sm
from tests.test_gpu_parity import _reshape_weights, K144_CONVS, _run_conv_hook
dev = torch.device("cuda:0")
mcfg = sm.default_cfg()
for dist in ["seeded", "student_t2", "rows_4dec", "rows_6dec", "chan_6dec", "x1e-3", "x1e3"]:
    p = sm.init_params(mcfg, seed=1)
    if dist != "seeded":
        p = _reshape_weights(p, dist, seed=77)
    model = dba.TensorProductModelHIP({}).to(dev); model.load_state_dict(p, strict=True)
    lib, h = L.load(), model.handle(dev)
    fb = model.fallback_convs(dev)
    p64 = {k: v.double() for k, v in p.items()}
    E, N = 600, 64
    for layer, fam, name in K144_CONVS:
        i, shirr, o, nef = sm.conv_specs(mcfg)[name]
        Din, Dout = o3.Irreps(i).dim, o3.Irreps(o).dim
        g = torch.Generator().manual_seed(11 + abs(layer))
        x, xt = torch.randn(N, Din, generator=g), torch.randn(N, max(Din, 48), generator=g)
        tgt = torch.sort(torch.randint(0, N, (E,), generator=g)).values
        gth = torch.randint(0, N, (E,), generator=g)
        emb = torch.randn(E, 48, generator=g)
        if "tor" in name:
            sh_full = torch.randn(E, o3.Irreps(shirr).dim, generator=g); sh9 = torch.zeros(E, 9); sh9[:, :7] = sh_full[:, :7]
        else:
            sh_full = o3.spherical_harmonics(shirr, torch.randn(E, 3, generator=g), True, "component"); sh9 = sh_full.clone()
        a64 = torch.cat([emb, xt[tgt, :48], x[gth, :48]], -1).double()
        m64 = sm._tp(i, shirr, o)(x[gth].double(), sh_full.double(), sm.simple_linear(p64, f"{name}.fc", a64))
        s64 = torch.zeros(N, Dout, dtype=torch.float64).index_add_(0, tgt, m64)
        c = dict(x=x.to(dev), xt=xt.to(dev), tgt=tgt.to(dev, torch.int32), gth=gth.to(dev, torch.int32), emb=emb.to(dev), sh=sh9.contiguous().to(dev), Din=Din, Dout=Dout)
        col = s64.abs().amax(dim=0).clamp_min(1e-300)
        res = {}
        for mode, fn in (("f32", lib.dbfr_test_conv), ("split_f16", lib.dbfr_test_conv2), ("reduce_first", lib.dbfr_test_conv2)):
            model.set_gemm(mode)
            m = _run_conv_hook(fn, h, layer, fam, c, E, dev).cpu()
            s = torch.zeros(N, Dout, dtype=torch.float64).index_add_(0, tgt, m.double())
            dm = (s - s64) / col
            res[mode] = (float(dm.abs().max()), float(dm.pow(2).mean().sqrt()), bool(torch.isfinite(m).all()))
        print(f"{dist:10s} {name:24s} seg-sum err vs f64 (max / rms per column): f32 {res['f32'][0]:.2e} / {res['f32'][1]:.2e}   split_f16 {res['split_f16'][0]:.2e} / {res['split_f16'][1]:.2e}   "
              f"reduce_first {res['reduce_first'][0]:.2e} / {res['reduce_first'][1]:.2e} finite={res['reduce_first'][2]}" + ("  [fallback]" if name in fb else ""), flush=True)
    model.release()
