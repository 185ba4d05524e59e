"""Developer probe of k_convz: one non-zero input channel at a time, ratio of segment sums to the oracle's."""
import ctypes as C, os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import diffbindfr_amd as dba
from diffbindfr_amd import lib as L
from oracle import e3nn_lite as o3, score_model as sm
dev = torch.device("cuda:0")
mcfg = sm.default_cfg(); p = sm.init_params(mcfg, seed=1)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "biasonly":
    for k in list(p):
        if k.endswith("fc.lin.3.weight"): p[k] = torch.zeros_like(p[k])
if mode == "nobias":
    for k in list(p):
        if k.endswith("fc.lin.3.bias"): p[k] = torch.zeros_like(p[k])
model = dba.TensorProductModelHIP({}).to(dev); model.load_state_dict(p, strict=True)
lib, h = L.load(), model.handle(dev)
ptr = lambda t: C.c_void_p(t.data_ptr())
name, layer, fam = "atom_conv_layers.3", 3, 2
i, shirr, o, nef = sm.conv_specs(mcfg)[name]
Din, Dout = o3.Irreps(i).dim, o3.Irreps(o).dim
g = torch.Generator().manual_seed(3)
E, Nt, Ng = 40, 3, 9
tgt = torch.sort(torch.randint(0, Nt, (E,), generator=g)).values
gth = torch.randint(0, Ng, (E,), generator=g)
emb = torch.randn(E, 48, generator=g); xt = torch.randn(Nt, Din, generator=g)
var = sys.argv[2] if len(sys.argv) > 2 else ""
if "xt" in var: xt[:] = xt[0]
if "emb" in var: emb[:] = emb[0]
if "tail" in var: emb[:16] = emb[0]
if "head" in var: emb[16:] = emb[16]
if var.startswith("one"):
    k = int(var[3:]); keep = emb[k].clone(); emb[:] = emb[0]; emb[k] = keep; xt[:] = xt[0]
sh = o3.spherical_harmonics(shirr, torch.randn(E, 3, generator=g), True, "component")
print("tgt", tgt.tolist())
for chans in ([0],):
    x = torch.zeros(Ng, Din); x[:, chans] = torch.randn(Ng, len(chans), generator=g) if len(chans) > 3 else 1.0
    a = torch.cat([emb, xt[tgt, :48], x[gth, :48]], -1)
    m_ref = sm._tp(i, shirr, o)(x[gth], sh, sm.simple_linear(p, f"{name}.fc", a))
    model.set_gemm("reduce_first")
    xd, xtd, embd, shd = x.to(dev), xt.to(dev), emb.to(dev), sh.contiguous().to(dev)
    tgtd, gthd = tgt.to(dev, torch.int32), gth.to(dev, torch.int32)
    ned = torch.tensor([E], dtype=torch.int32, device=dev)
    msg = torch.full((E, Dout), float("nan"), device=dev)
    L.check(lib.dbfr_test_conv2(h, layer, fam, E, ptr(ned), ptr(tgtd), ptr(gthd), ptr(embd), ptr(shd), ptr(xtd), xtd.shape[1], ptr(tgtd), ptr(xd), Din, ptr(gthd), ptr(xd), Din, ptr(msg), None))
    torch.cuda.synchronize(); msg = msg.cpu()
    s_ref = torch.zeros(Nt, Dout).index_add_(0, tgt, m_ref); s = torch.zeros(Nt, Dout).index_add_(0, tgt, msg)
    for io, sl in (("0e", slice(0, 48)), ("0o", slice(120, 168))):
        r, q = s_ref[:, sl], s[:, sl]
        print("   per-target err", [(float((q[t]-r[t]).abs().max())) for t in range(Nt)], " msg rows nonzero:", (msg[:, sl].abs().sum(1) > 0).nonzero().flatten().tolist()[:12])
        print(f"{mode} chans {chans[:3]}{'...' if len(chans)>3 else ''} {io}: max|ref| {float(r.abs().max()):.3e} max|got| {float(q.abs().max()):.3e} err {float((q-r).abs().max()):.3e}  ratio[t0,:6] {(q[0,:6]/r[0,:6]).tolist() if float(r.abs().max())>0 else '-'}")

if os.environ.get("DBFR_CONVZ_DEBUG"):
    W1 = p[f"{name}.fc.lin.0.weight"]; b1 = p[f"{name}.fc.lin.0.bias"]
    a = torch.cat([emb, xt[tgt, :48], x[gth, :48]], -1)
    href = torch.relu(a @ W1.T + b1)[:32]
    torch.save(href, os.environ["DBFR_CONVZ_DEBUG"] + ".ref.pt")
