"""Developer script (GPU box): unit check of the fused conv + reduce kernels vs the oracle."""
import sys

import torch

sys.path.insert(0, ".")
import ctypes as C
import diffbindfr_amd as dba
from diffbindfr_amd import lib as L
from oracle import score_model as sm
from oracle import e3nn_lite as o3
from tests.helpers import rel_err

dev = torch.device("cuda:0")
mcfg = sm.default_cfg()
p = sm.init_params(mcfg, seed=1)
model = dba.TensorProductModelHIP({}).to(dev)
model.load_state_dict(p, strict=True)
lib = L.load()
h = model.handle()
g = torch.Generator().manual_seed(5)
dims = [48, 84, 120, 168]
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for layer, fam, name in [(0, 0, "lig_conv_layers.0"), (1, 2, "atom_conv_layers.1"), (2, 1, "cross_al_conv_layers.2"),
                         (5, 3, "cross_la_conv_layers.5"), (-1, 0, "final_conv"), (-2, 0, "tor_bond_conv")]:
    i, shirr, o, nef = sm.conv_specs(mcfg)[name]
    Din, Dout = o3.Irreps(i).dim, o3.Irreps(o).dim
    Nt, Ng, E = 13, 17, 77
    x = torch.randn(Ng, Din, generator=g)
    xt = torch.randn(Nt, max(Din, 48), generator=g)
    tgt = torch.sort(torch.randint(0, Nt, (E,), generator=g)).values
    gth = torch.randint(0, Ng, (E,), generator=g)
    emb = torch.randn(E, 48, generator=g)
    shdim = o3.Irreps(shirr).dim
    vec = torch.randn(E, 3, generator=g)
    if name == "tor_bond_conv":
        sh_full = torch.randn(E, shdim, generator=g)
        sh9 = torch.zeros(E, 9)
        sh9[:, :7] = sh_full[:, :7]
    else:
        sh_full = o3.spherical_harmonics(shirr, vec, True, "component")
        sh9 = sh_full.clone()
    if nef == 144:
        a = torch.cat([emb, xt[tgt, :48], x[gth, :48]], -1)
    else:
        a = torch.cat([emb, x[gth, :48]], -1)
    w = sm.simple_linear(p, f"{name}.fc", a)
    m_ref = sm._tp(i, shirr, o)(x[gth], sh_full, w)
    # device
    D = lambda t, dt=None: t.to(dev) if dt is None else t.to(dev, dt)
    xd, xtd, embd, shd = D(x), D(xt), D(emb), D(sh9.contiguous())
    tgtd, gthd = D(tgt, torch.int32), D(gth, torch.int32)
    ned = torch.tensor([E], dtype=torch.int32, device=dev)
    msg = torch.zeros(E, Dout, device=dev)
    if nef == 144:
        rc = lib.dbfr_test_conv(h, layer, fam, E, ptr(ned), ptr(tgtd), ptr(gthd), ptr(embd), ptr(shd), ptr(xtd), xtd.shape[1],
                                ptr(tgtd), ptr(xd), Din, ptr(gthd), ptr(xd), Din, ptr(msg), None)
    else:
        rc = lib.dbfr_test_conv(h, layer, fam, E, ptr(ned), ptr(tgtd), ptr(gthd), ptr(embd), ptr(shd), ptr(xd), Din,
                                ptr(gthd), None, 0, ptr(gthd), ptr(xd), Din, ptr(msg), None)
    L.check(rc)
    torch.cuda.synchronize()
    print(f"{name:28s} msg rel err {rel_err(msg, m_ref):.3e}", end="  ")
    ir = o3.Irreps(o)
    off = 0
    for mi in ir:
        sl = slice(off, off + mi.dim)
        off += mi.dim
        print(f"[{mi.mul}x{mi.ir}: {rel_err(msg[:, sl], m_ref[:, sl]):.1e}]", end=" ")
    print()
    # reduce + LN
    from oracle.cluster import scatter
    out_ref = sm.layer_norm(p, f"{name}.batch_norm", o, scatter(m_ref, tgt, 0, Nt, "mean"))
    cnt = torch.bincount(tgt, minlength=Nt)
    rs = torch.cumsum(cnt, 0) - cnt
    outd = torch.zeros(Nt, Dout, device=dev)
    msg_ref_d = D(m_ref.contiguous())
    L.check(lib.dbfr_test_reduce_ln(h, layer, fam, ptr(msg_ref_d), ptr(D(rs, torch.int32)), ptr(D(cnt, torch.int32)), Nt, None, 0,
                                    ptr(outd), 2, None))
    torch.cuda.synchronize()
    print(f"{'':28s} reduce+LN rel err {rel_err(outd, out_ref):.3e}")
