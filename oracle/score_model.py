"""CPU restatement of the SE(3) score network ``TensorProductModel.forward``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, op for op,
druglib/models/Docking/interaction/tpscore.py (line numbers cited per function)
with the helper modules it pulls in:
  schnet.py:142-179 (GaussianSmearing), equibind_encoder.py:10-88 (AtomEncoder),
  Base/diffusion/time_emb.py:9-26 (sinusoidal_embedding),
  torch_utils/graph.py:81-140 (get_complete_bipartite_graph).
Parameters live in a flat ``dict`` keyed by the reference's own ``state_dict``
names (relative to ``TensorProductModel``), so the same
dict loads into the reference module, this oracle and the HIP product.

Materialises the per-edge weights ``[E, W]`` exactly like the reference does
(that is the CPU baseline that gets timed).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import e3nn_lite as o3
from .cluster import radius, radius_graph, scatter

# druglib/utils/obj/protein_constants.py:600 (atom_order of the 37 atom types)
ATOM37_CA, ATOM37_CB = 1, 3


def default_cfg(**over):
    """DiffBindFR/configs/diffbindfr_ts.py:107-142."""
    cfg = dict(
        task="struct_gen", no_sc_torsion=False,
        lig_node_features=27, lig_edge_features=10,
        atom_feature_dims=(37, 22, 4, 21, 2), atom_scalar_feats=0,
        ns=48, nv=12, sh_lmax=2, lig_cutoff=5.0, atom_cutoff=4.0, cross_cutoff=32.0,
        dynamic_max_cross=True, center_max_distance=32.0, atom_max_neighbors=1000,
        distance_embed_dim=32, sigma_embed_dim=32, emb_scale=1000, num_conv_layers=6,
        use_second_order_repr=False, batch_norm=True, scale_by_sigma=True,
    )
    cfg.update(over)
    return SimpleNamespace(**cfg)


def irrep_seq(cfg):
    """tpscore.py:283-288 (use_second_order_repr=False)."""
    ns, nv = cfg.ns, cfg.nv
    return [f"{ns}x0e", f"{ns}x0e + {nv}x1o", f"{ns}x0e + {nv}x1o + {nv}x1e",
            f"{ns}x0e + {nv}x1o + {nv}x1e + {ns}x0o"]


def conv_specs(cfg):
    """name -> (in_irreps, sh_irreps, out_irreps, n_edge_features) for the 27 convs
    (tpscore.py:293-308, 348-356, 374-382, 396-404)."""
    seq = irrep_seq(cfg)
    sh = o3.Irreps.spherical_harmonics(cfg.sh_lmax)
    ns = cfg.ns
    specs = {}
    for l in range(cfg.num_conv_layers):
        i, o = seq[min(l, 3)], seq[min(l + 1, 3)]
        for fam in ("lig", "atom", "cross_al", "cross_la"):
            specs[f"{fam}_conv_layers.{l}"] = (i, sh, o, 3 * ns)
    last = seq[min(cfg.num_conv_layers, 3)]
    specs["final_conv"] = (last, sh, "2x1o + 2x1e", 2 * ns)
    tor_sh = o3.FullTensorProduct(sh, "2e").irreps_out
    specs["tor_bond_conv"] = (last, tor_sh, f"{ns}x0o + {ns}x0e", 3 * ns)
    if not cfg.no_sc_torsion:
        specs["sc_tor_bond_conv"] = (last, tor_sh, f"{ns}x0o + {ns}x0e", 3 * ns)
    return specs


def param_shapes(cfg):
    """Every parameter/buffer of TensorProductModel with its shape (Appendix B.3)."""
    ns, se, de = cfg.ns, cfg.sigma_embed_dim, cfg.distance_embed_dim
    shp = {}

    def simple(name, i, o, h=None, bias=True):
        h = h or o
        shp[f"{name}.lin.0.weight"] = (h, i)
        shp[f"{name}.lin.3.weight"] = (o, h)
        if bias:
            shp[f"{name}.lin.0.bias"] = (h,)
            shp[f"{name}.lin.3.bias"] = (o,)

    simple("lig_node_embedding", cfg.lig_node_features + se, ns)
    simple("lig_edge_embedding", cfg.lig_edge_features + se + de, ns)
    for i, d in enumerate(cfg.atom_feature_dims):
        shp[f"atom_node_embedding.atom_emb_list.{i}.weight"] = (d, ns)
    shp["atom_node_embedding.scalar_lin.weight"] = (ns, cfg.atom_scalar_feats + se + ns)
    simple("atom_edge_embedding", se + de, ns)
    simple("la_edge_embedding", se + de, ns)
    for nm in ("lig", "atom", "cross", "center"):
        shp[f"{nm}_distance_expansion.coeff"] = ()
        shp[f"{nm}_distance_expansion.offset"] = (de,)
    for name, (i, sh, o, nef) in conv_specs(cfg).items():
        tp = o3.FullyConnectedTensorProduct(i, sh, o)
        simple(f"{name}.fc", nef, tp.weight_numel, nef)
        oi = o3.Irreps(o)
        shp[f"{name}.batch_norm.mean_shift"] = (1, oi.num_irreps, 1)
        shp[f"{name}.batch_norm.affine_weight"] = (oi.num_irreps,)
        shp[f"{name}.batch_norm.affine_bias"] = (sum(m.mul for m in oi if m.ir == (0, 1)),)
    simple("center_edge_embedding", de + se, ns)
    simple("tr_final_layer", 1 + se, 1, ns)
    simple("rot_final_layer", 1 + se, 1, ns)
    simple("tor_edge_embedding", de, ns)
    simple("tor_final_layer", 2 * ns, 1, ns, bias=False)
    if not cfg.no_sc_torsion:
        simple("sc_edge_embedding", de, ns)
        simple("sc_tor_final_layer", 2 * ns, 1, ns, bias=False)
    return shp


def init_params(cfg, seed=0):
    """Seeded random parameters (no checkpoint is available offline).  Linear
    weights ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's default, the
    LayerNorm parameters are perturbed away from their (1, 0) init so that a
    kernel ignoring them cannot pass parity."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    stops = dict(lig=cfg.lig_cutoff, atom=cfg.atom_cutoff, cross=cfg.cross_cutoff,
                 center=cfg.center_max_distance)
    for k, s in param_shapes(cfg).items():
        if k.endswith("distance_expansion.offset"):
            p[k] = torch.linspace(0.0, stops[k.split("_")[0]], cfg.distance_embed_dim)
        elif k.endswith("distance_expansion.coeff"):
            off = torch.linspace(0.0, stops[k.split("_")[0]], cfg.distance_embed_dim)
            p[k] = -0.5 / (off[1] - off[0]) ** 2
        elif k.endswith("mean_shift"):
            name = k[: -len(".batch_norm.mean_shift")]
            oi = o3.Irreps(conv_specs(cfg)[name][2])
            base = torch.cat([torch.ones(m.mul) if m.ir == (0, 1) else torch.zeros(m.mul) for m in oi])
            p[k] = (base + 0.1 * torch.randn(base.shape, generator=g)).view(1, -1, 1)
        elif k.endswith("affine_weight"):
            p[k] = 1.0 + 0.1 * torch.randn(s, generator=g)
        elif k.endswith("affine_bias"):
            p[k] = 0.1 * torch.randn(s, generator=g)
        elif "atom_emb_list" in k:
            p[k] = torch.randn(s, generator=g) * 0.5
        elif k.endswith(".weight"):
            bound = 1.0 / math.sqrt(s[1])
            p[k] = (torch.rand(s, generator=g) * 2 - 1) * bound
        elif k.endswith(".bias"):
            p[k] = (torch.rand(s, generator=g) * 2 - 1) * 0.1
        else:
            raise KeyError(k)
    return {k: v.float().contiguous() for k, v in p.items()}


# --------------------------------------------------------------------------- small modules
def simple_linear(p, name, x, act="relu"):
    """tpscore.py:109-141: Linear -> act -> (dropout off) -> Linear."""
    h = F.linear(x, p[f"{name}.lin.0.weight"], p.get(f"{name}.lin.0.bias"))
    h = torch.relu(h) if act == "relu" else torch.tanh(h)
    return F.linear(h, p[f"{name}.lin.3.weight"], p.get(f"{name}.lin.3.bias"))


def gaussian_smearing(p, name, dist):
    """schnet.py:174-179; clamp to `stop` (= offset[-1]) then expand."""
    off = p[f"{name}.offset"]
    d = dist.clamp_max(float(off[-1]))
    d = d.unsqueeze(-1) - off
    return torch.exp(p[f"{name}.coeff"] * torch.pow(d, 2))


def sinusoidal_embedding(timesteps, embed_dim, max_positions=10000):
    """Base/diffusion/time_emb.py:9-26."""
    half = embed_dim // 2
    emb = math.log(max_positions) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=timesteps.dtype) * -emb)
    emb = timesteps[:, None] * emb[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)


def atom_encoder(p, cfg, x):
    """equibind_encoder.py:68-88 (5 categorical embeddings + Linear(80->48, no bias))."""
    n1h = len(cfg.atom_feature_dims)
    x_emb = 0
    for i in range(n1h):
        x_emb = x_emb + p[f"atom_node_embedding.atom_emb_list.{i}.weight"][x[:, i].long()]
    scalar_dim = cfg.atom_scalar_feats + cfg.sigma_embed_dim
    x_emb = x_emb + F.linear(torch.cat([x_emb, x[:, n1h:n1h + scalar_dim]], dim=-1),
                             p["atom_node_embedding.scalar_lin.weight"])
    return x_emb


def layer_norm(p, name, irreps, x, eps=1e-5):
    """tpscore.py:53-104 (equivariant LayerNorm, 'component', learnable mean shift)."""
    ms, aw, ab = (p[f"{name}.mean_shift"], p[f"{name}.affine_weight"], p[f"{name}.affine_bias"])
    fields, ix, iw, ib, im = [], 0, 0, 0, 0
    for mi in o3.Irreps(irreps):
        mul, d = mi.mul, mi.ir.dim
        f = x.narrow(1, ix, mul * d).reshape(-1, mul, d)
        ix += mul * d
        f = f - f.mean(dim=1, keepdim=True).expand(-1, mul, -1) * ms.narrow(1, im, mul)
        im += mul
        n = f.pow(2).mean(-1).mean(dim=1, keepdim=True)
        n = (n + eps).pow(-0.5) * aw[None, iw:iw + mul]
        iw += mul
        f = f * n.reshape(-1, mul, 1)
        if d == 1 and mi.ir.p == 1:
            f = f + ab[ib:ib + mul].reshape(mul, 1)
            ib += mul
        fields.append(f.reshape(-1, mul * d))
    assert ix == x.shape[-1]
    return torch.cat(fields, dim=-1)


_TP_CACHE = {}


def _tp(i, sh, o):
    key = (str(i), str(sh), str(o))
    if key not in _TP_CACHE:
        _TP_CACHE[key] = o3.FullyConnectedTensorProduct(i, sh, o)
    return _TP_CACHE[key]


def tp_conv(p, cfg, name, node_attr, edge_index, edge_attr, edge_sh, out_nodes=None):
    """tpscore.py:177-199 (residual=False, batch_norm=True, reduce='mean')."""
    i, sh, o, _ = conv_specs(cfg)[name]
    edge_src, edge_dst = edge_index
    w = simple_linear(p, f"{name}.fc", edge_attr)          # [E, W] materialised (as the reference)
    m = _tp(i, sh, o)(node_attr[edge_dst], edge_sh, w)
    out_nodes = out_nodes or node_attr.shape[0]
    out = scatter(m, edge_src, dim=0, dim_size=out_nodes, reduce="mean")
    return layer_norm(p, f"{name}.batch_norm", o, out)


def complete_bipartite(n_src, n_dst):
    """torch_utils/graph.py:81-140, same edge order (graph, src-major, dst-minor)."""
    src_l, dst_l = [], []
    so, do = 0, 0
    for a, b in zip(n_src.tolist(), n_dst.tolist()):
        s = torch.arange(a).repeat_interleave(b) + so
        d = torch.arange(b).repeat(a) + do
        src_l.append(s)
        dst_l.append(d)
        so += a
        do += b
    return torch.stack([torch.cat(src_l), torch.cat(dst_l)]).long()


# --------------------------------------------------------------------------- graphs
def _sh(cfg, vec):
    return o3.spherical_harmonics(o3.Irreps.spherical_harmonics(cfg.sh_lmax), vec,
                                  normalize=True, normalization="component")


def build_lig_conv_graph(p, cfg, data):
    """tpscore.py:575-600."""
    data.lig_node_sigma_emb = data.time_emb[data.lig_node_batch]
    node_attr = torch.cat([data.lig_node, data.lig_node_sigma_emb], 1)
    radius_edges = radius_graph(data.lig_pos, cfg.lig_cutoff, data.lig_node_batch)
    edge_index = torch.cat([data.lig_edge_index, radius_edges], 1).long()
    edge_attr = torch.cat([data.lig_edge_feat,
                           torch.zeros(radius_edges.shape[-1], cfg.lig_edge_features)], 0)
    src, dst = edge_index
    edge_sigma_emb = data.lig_node_sigma_emb[src]
    edge_vec = data.lig_pos[dst] - data.lig_pos[src]
    edge_length_emb = gaussian_smearing(p, "lig_distance_expansion", edge_vec.norm(dim=-1))
    edge_attr = torch.cat([edge_attr, edge_sigma_emb, edge_length_emb], 1)
    return node_attr, edge_index, edge_attr, _sh(cfg, edge_vec)


def build_atom_conv_graph(p, cfg, data):
    """tpscore.py:602-622."""
    emb = data.time_emb[data.rec_atm_pos_batch]
    node_attr = torch.cat([data.pocket_node_feature, emb], 1)
    edge_index = radius_graph(data.rec_atm_pos, cfg.atom_cutoff, data.rec_atm_pos_batch,
                              max_num_neighbors=cfg.atom_max_neighbors)
    src, dst = edge_index
    edge_vec = data.rec_atm_pos[dst] - data.rec_atm_pos[src]
    edge_length_emb = gaussian_smearing(p, "atom_distance_expansion", edge_vec.norm(dim=-1))
    edge_attr = torch.cat([emb[src], edge_length_emb], 1)
    return node_attr, edge_index, edge_attr, _sh(cfg, edge_vec)


def cross_edge_index(cfg, data):
    """tpscore.py:624-664: lig x {CA,CB} complete bipartite (+) radius on the other
    atoms with the per-graph cutoff 0.2*tr_sigma+5 realised by coordinate scaling."""
    atom37_id = data.pocket_node_feature[:, 0].long()
    cab_mask = (atom37_id == ATOM37_CA) | (atom37_id == ATOM37_CB)
    atomids = torch.arange(data.pocket_node_feature.size(0))
    cab_indx = atomids[cab_mask]
    cab_bincount = torch.bincount(data.rec_atm_pos_batch[cab_indx])
    lig_bincount = torch.bincount(data.lig_node_batch)
    lrab = complete_bipartite(lig_bincount, cab_bincount)
    lrab = torch.stack([lrab[0], cab_indx[lrab[1]]], dim=0)
    nab_mask = ~cab_mask
    nab_indx = atomids[nab_mask]
    nab_batch = data.rec_atm_pos_batch[nab_indx]
    nab_pos = data.rec_atm_pos[nab_mask]
    if cfg.dynamic_max_cross:
        cut = data.tr_sigma * 0.2 + 5          # [G, 1]
        lnab = radius(nab_pos / cut[nab_batch], data.lig_pos / cut[data.lig_node_batch], 1,
                      nab_batch, data.lig_node_batch, max_num_neighbors=10000)
    else:
        lnab = radius(nab_pos, data.lig_pos, cfg.cross_cutoff, nab_batch, data.lig_node_batch,
                      max_num_neighbors=10000)
    lnab = torch.stack([lnab[0], nab_indx[lnab[1]]], dim=0)
    return torch.cat([lrab, lnab], dim=1).long()


def build_cross_conv_graph(p, cfg, data):
    """tpscore.py:666-682."""
    ei = cross_edge_index(cfg, data)
    vec = data.rec_atm_pos[ei[1]] - data.lig_pos[ei[0]]
    length_emb = gaussian_smearing(p, "cross_distance_expansion", vec.norm(dim=-1))
    attr = torch.cat([data.lig_node_sigma_emb[ei[0]], length_emb], 1)
    return ei, attr, _sh(cfg, vec)


def build_center_conv_graph(p, cfg, data):
    """tpscore.py:684-710."""
    n = len(data.lig_node_batch)
    edge_index = torch.stack([data.lig_node_batch, torch.arange(n)], 0).long()
    center = torch.zeros(data.num_graphs, 3)
    center.index_add_(0, data.lig_node_batch, data.lig_pos)
    center = center / torch.bincount(data.lig_node_batch).unsqueeze(1)
    vec = data.lig_pos[edge_index[1]] - center[edge_index[0]]
    length_emb = gaussian_smearing(p, "center_distance_expansion", vec.norm(dim=-1))
    attr = torch.cat([data.lig_node_sigma_emb[edge_index[1]], length_emb], 1)
    return edge_index, attr, _sh(cfg, vec)


_FTP = {}


def _final_tp_tor(cfg):
    if cfg.sh_lmax not in _FTP:
        _FTP[cfg.sh_lmax] = o3.FullTensorProduct(o3.Irreps.spherical_harmonics(cfg.sh_lmax), "2e")
    return _FTP[cfg.sh_lmax]


def _bond_conv_graph(p, cfg, pos, pos_batch, bonds, node_attr, cutoff, dist_name, emb_name):
    """Shared body of tpscore.py:712-734 and :736-759 (pseudotorque graphs)."""
    bond_vec = pos[bonds[1]] - pos[bonds[0]]
    bond_attr = node_attr[bonds[0]] + node_attr[bonds[1]]
    bond_sh = o3.spherical_harmonics("2e", bond_vec, normalize=True, normalization="component")
    bond_pos = (pos[bonds[0]] + pos[bonds[1]]) / 2
    bond_batch = pos_batch[bonds[0]]
    edge_index = radius(pos, bond_pos, cutoff, batch_x=pos_batch, batch_y=bond_batch)
    vec = pos[edge_index[1]] - bond_pos[edge_index[0]]
    attr = simple_linear(p, emb_name, gaussian_smearing(p, dist_name, vec.norm(dim=-1)))
    sh = _final_tp_tor(cfg)(_sh(cfg, vec), bond_sh[edge_index[0]])
    ns = cfg.ns
    attr = torch.cat([attr, node_attr[edge_index[1], :ns], bond_attr[edge_index[0], :ns]], -1)
    return edge_index, attr, sh


# --------------------------------------------------------------------------- forward
def forward(p, cfg, data):
    """tpscore.py:462-573.  ``data`` is an attribute namespace holding the batched
    tensors of SURVEY.md Appendix B.1 plus ``t``, ``tr_sigma`` [G], ``rot_score_norm``
    [G,1], ``tor_score_norm2`` [n_tor], ``sc_tor_score_norm2`` [N_res,4].  Mutated in
    place exactly where the reference mutates its EasyDict."""
    ns = cfg.ns
    data.num_graphs = int(data.lig_node_batch.max().item()) + 1
    data.time_emb = sinusoidal_embedding(cfg.emb_scale * data.t, cfg.sigma_embed_dim)

    lig_node_attr, lig_ei, lig_edge_attr, lig_sh = build_lig_conv_graph(p, cfg, data)
    lig_node_attr = simple_linear(p, "lig_node_embedding", lig_node_attr)
    lig_edge_attr = simple_linear(p, "lig_edge_embedding", lig_edge_attr)

    atom_node_attr, atom_ei, atom_edge_attr, atom_sh = build_atom_conv_graph(p, cfg, data)
    atom_node_attr = atom_encoder(p, cfg, atom_node_attr)
    atom_edge_attr = simple_linear(p, "atom_edge_embedding", atom_edge_attr)

    data.tr_sigma = data.tr_sigma.unsqueeze(1)
    la_ei, la_edge_attr, la_sh = build_cross_conv_graph(p, cfg, data)
    la_edge_attr = simple_linear(p, "la_edge_embedding", la_edge_attr)

    if not cfg.no_sc_torsion:
        data.sc_torsion_edge_index = data.torsion_edge_index[data.sc_torsion_edge_mask].T

    for l in range(cfg.num_conv_layers):
        a = torch.cat([lig_edge_attr, lig_node_attr[lig_ei[0], :ns], lig_node_attr[lig_ei[1], :ns]], -1)
        lig_update = tp_conv(p, cfg, f"lig_conv_layers.{l}", lig_node_attr, lig_ei, a, lig_sh)

        a = torch.cat([la_edge_attr, lig_node_attr[la_ei[0], :ns], atom_node_attr[la_ei[1], :ns]], -1)
        al_update = tp_conv(p, cfg, f"cross_al_conv_layers.{l}", atom_node_attr, la_ei, a, la_sh,
                            out_nodes=lig_node_attr.shape[0])

        a = torch.cat([atom_edge_attr, atom_node_attr[atom_ei[0], :ns], atom_node_attr[atom_ei[1], :ns]], -1)
        atom_update = tp_conv(p, cfg, f"atom_conv_layers.{l}", atom_node_attr, atom_ei, a, atom_sh)

        a = torch.cat([la_edge_attr, atom_node_attr[la_ei[1], :ns], lig_node_attr[la_ei[0], :ns]], -1)
        la_update = tp_conv(p, cfg, f"cross_la_conv_layers.{l}", lig_node_attr, torch.flip(la_ei, dims=[0]),
                            a, la_sh, out_nodes=atom_node_attr.shape[0])

        lig_node_attr = F.pad(lig_node_attr, (0, lig_update.shape[-1] - lig_node_attr.shape[-1]))
        lig_node_attr = lig_node_attr + lig_update + al_update
        atom_node_attr = F.pad(atom_node_attr, (0, atom_update.shape[-1] - atom_node_attr.shape[-1]))
        atom_node_attr = atom_node_attr + atom_update + la_update

    data._lig_node_attr, data._atom_node_attr = lig_node_attr, atom_node_attr  # debug taps

    c_ei, c_attr, c_sh = build_center_conv_graph(p, cfg, data)
    c_attr = simple_linear(p, "center_edge_embedding", c_attr)
    c_attr = torch.cat([c_attr, lig_node_attr[c_ei[1], :ns]], -1)
    global_pred = tp_conv(p, cfg, "final_conv", lig_node_attr, c_ei, c_attr, c_sh,
                          out_nodes=data.num_graphs)
    tr_pred = global_pred[:, :3] + global_pred[:, 6:9]
    rot_pred = global_pred[:, 3:6] + global_pred[:, 9:]

    tr_norm = torch.linalg.vector_norm(tr_pred, dim=1).unsqueeze(1)
    tr_pred = tr_pred / tr_norm * simple_linear(p, "tr_final_layer", torch.cat([tr_norm, data.time_emb], dim=1))
    rot_norm = torch.linalg.vector_norm(rot_pred, dim=1).unsqueeze(1)
    rot_pred = rot_pred / rot_norm * simple_linear(p, "rot_final_layer", torch.cat([rot_norm, data.time_emb], dim=1))

    n_tor = int(data.tor_edge_mask.sum())
    if n_tor > 0:
        bonds = data.lig_edge_index[:, data.tor_edge_mask.bool()]
        t_ei, t_attr, t_sh = _bond_conv_graph(p, cfg, data.lig_pos, data.lig_node_batch, bonds, lig_node_attr,
                                              cfg.lig_cutoff, "lig_distance_expansion", "tor_edge_embedding")
        tor_pred = tp_conv(p, cfg, "tor_bond_conv", lig_node_attr, t_ei, t_attr, t_sh, out_nodes=n_tor)
        tor_pred = simple_linear(p, "tor_final_layer", tor_pred, act="tanh").squeeze(1)
    else:
        tor_pred = torch.empty(0)

    if cfg.scale_by_sigma:
        tr_pred = tr_pred / data.tr_sigma
        rot_pred = rot_pred * data.rot_score_norm
        if n_tor > 0:
            tor_pred = tor_pred * torch.sqrt(data.tor_score_norm2)

    if cfg.no_sc_torsion:
        return tr_pred, rot_pred, tor_pred, None

    n_sc = int(data.sc_torsion_edge_mask.sum())
    s_ei, s_attr, s_sh = _bond_conv_graph(p, cfg, data.rec_atm_pos, data.rec_atm_pos_batch,
                                          data.sc_torsion_edge_index, atom_node_attr, cfg.atom_cutoff,
                                          "atom_distance_expansion", "sc_edge_embedding")
    sc_pred = tp_conv(p, cfg, "sc_tor_bond_conv", atom_node_attr, s_ei, s_attr, s_sh, out_nodes=n_sc)
    sc_pred = simple_linear(p, "sc_tor_final_layer", sc_pred, act="tanh").squeeze(1)
    data.sc_tor_score_norm2 = data.sc_tor_score_norm2[data.sc_torsion_edge_mask]
    if cfg.scale_by_sigma:
        sc_pred = sc_pred * torch.sqrt(data.sc_tor_score_norm2)
    return tr_pred, rot_pred, tor_pred, sc_pred
