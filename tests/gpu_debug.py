"""Developer script (GPU box): stage-by-stage comparison of the HIP internals vs the oracle."""
import copy
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from tests.helpers import load_golden_batch, namespace_to, rel_err  # noqa
import diffbindfr_amd as dba  # noqa
from diffbindfr_amd import lib as L  # noqa
from diffbindfr_amd.packing import PackedBatch  # noqa
from oracle import sampler as osampler, schedule as osched, score_model as sm  # noqa

dev = torch.device("cuda:0")
mcfg = sm.default_cfg()
p = sm.init_params(mcfg, seed=1)
model = dba.TensorProductModelHIP({}).to(dev)
model.load_state_dict(p, strict=True)
d, z = load_golden_batch()
G = d.num_graphs
scfg = osched.default_sample_cfg()
step = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc = osched.step_scalars(scfg, step)
dd = osampler.set_time(copy.deepcopy(d), sc, G)
dg = namespace_to(dd, dev)
pb = PackedBatch(dg, dev)
scn = dg.sc_tor_score_norm2[dg.sc_torsion_edge_mask]
out = model.score_packed(pb, dg.t, dg.tr_sigma, dg.rot_score_norm, dg.tor_score_norm2, scn)
ws = model._ws
V = L.workspace_views(model.handle(), pb.c, model.limits, ws)
f32 = lambda n: V[n].view(torch.float32).cpu()
i32 = lambda n: V[n].view(torch.int32).cpu()
ne = i32("n_edges")
print("n_edges [ll aa al la tor sc - center]:", ne.tolist())

# ---- oracle stages
data = copy.deepcopy(dd)
data.num_graphs = G
data.time_emb = sm.sinusoidal_embedding(mcfg.emb_scale * data.t, 32)
print("temb err", (f32("temb").view(G, 32) - data.time_emb).abs().max().item())
lig_node_attr, lig_ei, lig_edge_attr, lig_sh = sm.build_lig_conv_graph(p, mcfg, data)
lig_x = sm.simple_linear(p, "lig_node_embedding", lig_node_attr)
lig_e = sm.simple_linear(p, "lig_edge_embedding", lig_edge_attr)
atom_node_attr, atom_ei, atom_edge_attr, atom_sh = sm.build_atom_conv_graph(p, mcfg, data)
atom_x = sm.atom_encoder(p, mcfg, atom_node_attr)
atom_e = sm.simple_linear(p, "atom_edge_embedding", atom_edge_attr)
data.tr_sigma = data.tr_sigma.unsqueeze(1)
la_ei, la_edge_attr, la_sh = sm.build_cross_conv_graph(p, mcfg, data)
la_e = sm.simple_linear(p, "la_edge_embedding", la_edge_attr)
print("oracle edges ll/aa/la:", lig_ei.shape[1], atom_ei.shape[1], la_ei.shape[1])


def edge_dict(tgt, gth, vals):
    return {(int(a), int(b), i): v for i, (a, b, v) in enumerate(zip(tgt, gth, vals))}


def cmp_set(name, n, o_tgt, o_gth, o_sh, o_emb, dup_ok=False):
    tgt, gth = i32(name + ".tgt")[:n], i32(name + ".gth")[:n]
    sh = f32(name + ".sh").view(-1, 9)[:n]
    emb = f32(name + ".emb").view(-1, 48)[:n]
    # match edges by (tgt,gth) pair; duplicates (bond + radius) matched by sorted order of sh then emb norm
    from collections import defaultdict
    A, B = defaultdict(list), defaultdict(list)
    for i in range(n):
        A[(int(tgt[i]), int(gth[i]))].append(i)
    for i in range(o_tgt.shape[0]):
        B[(int(o_tgt[i]), int(o_gth[i]))].append(i)
    missing = [k for k in B if k not in A]
    extra = [k for k in A if k not in B]
    worst_sh = worst_emb = 0.0
    for k in B:
        if k in A and len(A[k]) == len(B[k]):
            ia = sorted(A[k], key=lambda i: float(emb[i].sum()))
            ib = sorted(B[k], key=lambda i: float(o_emb[i].sum()))
            for x, y in zip(ia, ib):
                worst_sh = max(worst_sh, float((sh[x] - o_sh[y]).abs().max()))
                worst_emb = max(worst_emb, float((emb[x] - o_emb[y]).abs().max()))
    print(f"  set {name}: n={n} oracle={o_tgt.shape[0]} missing={len(missing)} extra={len(extra)} sh_err={worst_sh:.2e} emb_err={worst_emb:.2e}")


cmp_set("ll", int(ne[0]), lig_ei[0], lig_ei[1], lig_sh, lig_e)
cmp_set("aa", int(ne[1]), atom_ei[0], atom_ei[1], atom_sh, atom_e)
cmp_set("al", int(ne[2]), la_ei[0], la_ei[1], la_sh, la_e)
cmp_set("la", int(ne[3]), la_ei[1], la_ei[0], la_sh, la_e)

# after the full forward lig_x0 etc. have been overwritten by later layers; the final features:
ns = 48
nl, na = lig_x.shape[0], atom_x.shape[0]
o_lig, o_atom = lig_x, atom_x
for l in range(mcfg.num_conv_layers):
    a = torch.cat([lig_e, o_lig[lig_ei[0], :ns], o_lig[lig_ei[1], :ns]], -1)
    lu = sm.tp_conv(p, mcfg, f"lig_conv_layers.{l}", o_lig, lig_ei, a, lig_sh)
    a = torch.cat([la_e, o_lig[la_ei[0], :ns], o_atom[la_ei[1], :ns]], -1)
    alu = sm.tp_conv(p, mcfg, f"cross_al_conv_layers.{l}", o_atom, la_ei, a, la_sh, out_nodes=nl)
    a = torch.cat([atom_e, o_atom[atom_ei[0], :ns], o_atom[atom_ei[1], :ns]], -1)
    au = sm.tp_conv(p, mcfg, f"atom_conv_layers.{l}", o_atom, atom_ei, a, atom_sh)
    a = torch.cat([la_e, o_atom[la_ei[1], :ns], o_lig[la_ei[0], :ns]], -1)
    lau = sm.tp_conv(p, mcfg, f"cross_la_conv_layers.{l}", o_lig, torch.flip(la_ei, dims=[0]), a, la_sh, out_nodes=na)
    o_lig = F.pad(o_lig, (0, lu.shape[-1] - o_lig.shape[-1])) + lu + alu
    o_atom = F.pad(o_atom, (0, au.shape[-1] - o_atom.shape[-1])) + au + lau
    if l == 0:
        first = (o_lig.clone(), o_atom.clone(), lu, alu)
# 6 layers => final features sit in buffer index 0 (6 swaps)
h_lig = f32("lig_x0").view(-1)[: nl * 168].view(nl, 168)
h_atom = f32("atom_x0").view(-1)[: na * 168].view(na, 168)
print("final lig feat rel err", rel_err(h_lig, o_lig), " atom", rel_err(h_atom, o_atom))
for nm, sl in (("0e", slice(0, 48)), ("1o", slice(48, 84)), ("1e", slice(84, 120)), ("0o", slice(120, 168))):
    print("   block", nm, "lig", rel_err(h_lig[:, sl], o_lig[:, sl]), "atom", rel_err(h_atom[:, sl], o_atom[:, sl]))
# layer-0 output sits in buffer 1 only if we stop after 1 layer; run a 1-layer model for that
print("scores rel err:", [rel_err(a, torch.from_numpy(z[f"score_{n}_{step}"])) for a, n in zip(out, ("tr", "rot", "tor", "sc_tor"))])
print("is_cab sum", int(V["is_cab"].cpu().sum()), "na", na)
print("la.row_cnt", i32("la.row_cnt")[:na].tolist()[:70])
print("la.g_cnt", i32("la.g_cnt").tolist(), "la.g_base", i32("la.g_base").tolist())
print("al.g_cnt", i32("al.g_cnt").tolist())
