"""Developer check of DBFR_GEMM_REDUCE_FIRST (convz.hip) on single convs through the C-ABI test hooks:
segment sums of the messages against the fp32-instruction kernel and the oracle's tensor product, then timing.
    python tools/exp/convz_check.py [--time]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import diffbindfr_amd as dba  # noqa: E402
from diffbindfr_amd import lib as L  # noqa: E402
from oracle import e3nn_lite as o3, score_model as sm  # noqa: E402

dev = torch.device("cuda:0")
mcfg = sm.default_cfg()
p = sm.init_params(mcfg, seed=1)
model = dba.TensorProductModelHIP({}).to(dev)
model.load_state_dict(p, strict=True)
lib, h = L.load(), model.handle(dev)
ptr = lambda t: C.c_void_p(t.data_ptr())


def inputs(name, E, Nt, Ng, seed=5, deg=None):
    g = torch.Generator().manual_seed(seed)
    i, shirr, o, nef = sm.conv_specs(mcfg)[name]
    Din, Dout = o3.Irreps(i).dim, o3.Irreps(o).dim
    x, xt = torch.randn(Ng, Din, generator=g), torch.randn(Nt, max(Din, 48), generator=g)
    tgt = torch.sort(torch.randint(0, Nt, (E,), generator=g)).values
    gth = torch.randint(0, Ng, (E,), generator=g)
    emb = torch.randn(E, 48, generator=g)
    if "tor" in name:
        sh_full = torch.randn(E, o3.Irreps(shirr).dim, generator=g)
        sh9 = torch.zeros(E, 9)
        sh9[:, :7] = sh_full[:, :7]
    else:
        sh_full = o3.spherical_harmonics(shirr, torch.randn(E, 3, generator=g), True, "component")
        sh9 = sh_full.clone()
    return dict(name=name, i=i, shirr=shirr, o=o, Din=Din, Dout=Dout, x=x, xt=xt, tgt=tgt, gth=gth, emb=emb, sh_full=sh_full, sh9=sh9, E=E, Nt=Nt)


def run(c, layer, fam, mode, fn):
    model.set_gemm(mode)
    E = c["E"]
    xd, xtd, embd, shd = c["x"].to(dev), c["xt"].to(dev), c["emb"].to(dev), c["sh9"].contiguous().to(dev)
    tgtd, gthd = c["tgt"].to(dev, torch.int32), c["gth"].to(dev, torch.int32)
    ned = torch.tensor([E], dtype=torch.int32, device=dev)
    msg = torch.full((E, c["Dout"]), float("nan"), device=dev)
    L.check(fn(h, layer, fam, E, ptr(ned), ptr(tgtd), ptr(gthd), ptr(embd), ptr(shd), ptr(xtd), xtd.shape[1], ptr(tgtd), ptr(xd), c["Din"],
               ptr(gthd), ptr(xd), c["Din"], ptr(msg), None))
    torch.cuda.synchronize()
    return msg.cpu()


def seg(m, tgt, Nt):
    out = torch.zeros(Nt, m.shape[1], dtype=m.dtype)
    return out.index_add_(0, tgt, m)


ok = True
for (layer, fam, name, E, Nt, Ng) in [] if "--timeonly" in sys.argv else [(3, 2, "atom_conv_layers.3", 77, 13, 17), (3, 2, "atom_conv_layers.3", 3000, 180, 200), (0, 0, "lig_conv_layers.0", 900, 60, 64),
                                      (1, 2, "atom_conv_layers.1", 1500, 400, 90), (2, 1, "cross_al_conv_layers.2", 2000, 40, 300), (5, 3, "cross_la_conv_layers.5", 2500, 900, 50),
                                      (-2, 0, "tor_bond_conv", 1200, 70, 80)]:
    c = inputs(name, E, Nt, Ng)
    a = torch.cat([c["emb"], c["xt"][c["tgt"], :48], c["x"][c["gth"], :48]], -1)
    m_ref = sm._tp(c["i"], c["shirr"], c["o"])(c["x"][c["gth"]], c["sh_full"], sm.simple_linear(p, f"{name}.fc", a))
    m32 = run(c, layer, fam, "f32", lib.dbfr_test_conv)
    mz = run(c, layer, fam, "reduce_first", lib.dbfr_test_conv2)
    s_ref, s32, sz = seg(m_ref, c["tgt"], Nt), seg(m32, c["tgt"], Nt), seg(mz, c["tgt"], Nt)
    den = float(s_ref.abs().max())
    e32, ez = float((s32 - s_ref).abs().max()) / den, float((sz - s_ref).abs().max()) / den
    fin = bool(torch.isfinite(mz).all())
    print(f"{name:26s} E={E:5d} targets={Nt:4d}: segment sums vs oracle: k_conv {e32:.2e}  reduce_first {ez:.2e}  finite={fin}", flush=True)
    if not fin or ez > 2e-5:
        ok = False
        bad = ((sz - s_ref).abs() / den > 2e-5)
        cols = bad.any(0).nonzero().flatten().tolist()
        rows = bad.any(1).nonzero().flatten().tolist()
        print("   bad columns", cols[:20], "... n =", len(cols), " bad rows", rows[:10], "n =", len(rows))
        nanc = (~torch.isfinite(mz)).any(0).nonzero().flatten().tolist()
        print("   non-finite columns", nanc[:20], "n =", len(nanc))
print("OK" if ok else "MISMATCH")

if "--time" in sys.argv or "--timeonly" in sys.argv:
    for (layer, fam, name, E, Nt, Ng) in [(3, 2, "atom_conv_layers.3", 650000, 50000, 65000), (3, 1, "cross_al_conv_layers.3", 650000, 13000, 65000), (0, 0, "lig_conv_layers.0", 650000, 45000, 45000)]:
        c = inputs(name, E, Nt, Ng)
        for mode in ("split_f16", "reduce_first"):
            model.set_gemm(mode)
            xd, xtd, embd, shd = c["x"].to(dev), c["xt"].to(dev), c["emb"].to(dev), c["sh9"].contiguous().to(dev)
            tgtd, gthd = c["tgt"].to(dev, torch.int32), c["gth"].to(dev, torch.int32)
            ned = torch.tensor([E], dtype=torch.int32, device=dev)
            msg = torch.zeros(E, c["Dout"], device=dev)
            call = lambda: L.check(lib.dbfr_test_conv2(h, layer, fam, E, ptr(ned), ptr(tgtd), ptr(gthd), ptr(embd), ptr(shd), ptr(xtd), xtd.shape[1], ptr(tgtd), ptr(xd), c["Din"],
                                                       ptr(gthd), ptr(xd), c["Din"], ptr(msg), None))
            call(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            print(f"{name} E={E} targets={Nt} {mode:13s}: {min(ts) * 1e3:.3f} ms", flush=True)
