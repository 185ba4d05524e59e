"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the MDN pose scorer's network forward.

Follows DiffBindFR/scoring/architecture/KarmaDock_sc.py:58-101 (`KarmaDock.forward` = `encoding` + `scoring`):
  ligand encoder    GraphTransformer_Block.py:56-88 (attention), :164-222 (layer), :314-352 (final layer), :406-424
  pocket encoder    GVP_Block.py:53-67 (GVP_embedding.forward), :163-190 (GVP), :235-244 (LayerNorm), :300-317 (GVPConv,
                    PyG MessagePassing with aggr='mean': messages j -> i, i = edge_index[1]), :431-466 (GVPConvLayer)
  mixture head      MDN_Block.py:21-66 (pair features, mixture parameters, min-over-atom14 distance), :74-79 (probability)
  score             KarmaDock_sc.py:98-101 (pairs beyond 5 A zeroed, summed per graph)
in eval mode (dropout off, BatchNorm1d with its running statistics).  Parameters are a flat dict keyed like the
reference module's state_dict (``lig_encoder.*``, ``pro_encoder.*``, ``mdn_layer.*``).

The third-party pieces the reference calls here -- torch_geometric ``MessagePassing.propagate`` / ``to_dense_batch``,
torch_scatter ``scatter`` / ``scatter_add`` -- are absent offline and restated from their documented semantics
(parity UNPINNED at that boundary, like e3nn); everything else is pinned by tests/golden/make_golden.py::golden_mdn, which
runs the reference's own architecture files on the same inputs (tests/golden/mdn.npz).
"""
import math

import torch
import torch.nn.functional as F

GT_LAYERS, GT_HEADS, GT_DIM = 6, 4, 128
GVP_LAYERS = 3


def _lin(p, k, x):
    b = p.get(k + ".bias")
    return F.linear(x, p[k + ".weight"], b)


def _bn(p, k, x, eps=1e-5):
    """nn.BatchNorm1d in eval mode."""
    return (x - p[k + ".running_mean"]) / torch.sqrt(p[k + ".running_var"] + eps) * p[k + ".weight"] + p[k + ".bias"]


def _scatter_add(src, index, n):
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add_(0, index, src)


# ------------------------------------------------------------------------------------------------ graph transformer
def _mha(p, k, x, e, edge_index, update_edges):
    """MultiHeadAttentionLayer (GraphTransformer_Block.py:56-88): H = 4 heads of 32."""
    H, d = GT_HEADS, GT_DIM // GT_HEADS
    q = _lin(p, k + ".Q", x).view(-1, H, d)
    kk = _lin(p, k + ".K", x).view(-1, H, d)
    v = _lin(p, k + ".V", x).view(-1, H, d)
    ep = _lin(p, k + ".edge_feats_projection", e).view(-1, H, d)
    row, col = edge_index
    alpha = kk[row] * q[col]
    alpha = (alpha / math.sqrt(d)).clamp(-5.0, 5.0)
    alpha = alpha * ep
    e_out = alpha if update_edges else None
    ax = torch.exp(alpha.sum(-1, keepdim=True).clamp(-5.0, 5.0))
    n = x.shape[0]
    wv = _scatter_add(v[row] * ax, col, n)
    z = _scatter_add(ax, col, n)
    return wv / (z + torch.full_like(z, 1e-6)), e_out


def _gt_layer(p, k, edge_index, x, e, final):
    x_in, e_in = x, e
    x = _bn(p, k + ".batch_norm1_node_feats", x)
    e = _bn(p, k + ".batch_norm1_edge_feats", e)
    x, e = _mha(p, k + ".mha_module", x, e, edge_index, not final)
    x = x_in + _lin(p, k + ".O_node_feats", x.reshape(-1, GT_DIM))
    x_in2 = x
    x = _bn(p, k + ".batch_norm2_node_feats", x)
    x = x_in2 + F.linear(F.silu(F.linear(x, p[k + ".node_feats_MLP.0.weight"])), p[k + ".node_feats_MLP.3.weight"])
    if final:
        return x, None
    e = e_in + _lin(p, k + ".O_edge_feats", e.reshape(-1, GT_DIM))
    e_in2 = e
    e = _bn(p, k + ".batch_norm2_edge_feats", e)
    e = e_in2 + F.linear(F.silu(F.linear(e, p[k + ".edge_feats_MLP.0.weight"])), p[k + ".edge_feats_MLP.3.weight"])
    return x, e


def ligand_encoder(p, node_s, edge_s, edge_index, prefix="lig_encoder"):
    x = _lin(p, prefix + ".node_encoder", node_s)
    e = _lin(p, prefix + ".edge_encoder", edge_s)
    for l in range(GT_LAYERS):
        x, e = _gt_layer(p, f"{prefix}.gt_block.{l}", edge_index, x, e, final=l == GT_LAYERS - 1)
    return x


# ------------------------------------------------------------------------------------------------ GVP
def _norm_no_nan(x, axis=-1, keepdims=False, eps=1e-8, sqrt=True):
    out = torch.clamp(torch.sum(torch.square(x), axis, keepdims), min=eps)
    return torch.sqrt(out) if sqrt else out


def _gvp(p, k, s, v, vo, scalar_act, vector_act):
    """GVP.forward (GVP_Block.py:163-190), vector_gate=False, vi > 0."""
    vt = torch.transpose(v, -1, -2)                     # [*, 3, vi]
    vh = F.linear(vt, p[k + ".wh.weight"])              # [*, 3, h]
    vn = _norm_no_nan(vh, axis=-2)                      # [*, h]
    s = _lin(p, k + ".ws", torch.cat([s, vn], -1))
    vout = None
    if vo:
        vout = torch.transpose(F.linear(vh, p[k + ".wv.weight"]), -1, -2)    # [*, vo, 3]
        if vector_act:
            vout = vout * torch.sigmoid(_norm_no_nan(vout, axis=-1, keepdims=True))
    if scalar_act:
        s = F.relu(s)
    return s, vout


def _gvp_ln(p, k, s, v):
    """LayerNorm for (s, V) tuples (GVP_Block.py:235-244)."""
    s = F.layer_norm(s, (s.shape[-1],), p[k + ".scalar_norm.weight"], p[k + ".scalar_norm.bias"])
    if v is None:
        return s, None
    vn = _norm_no_nan(v, axis=-1, keepdims=True, sqrt=False)
    vn = torch.sqrt(torch.mean(vn, dim=-2, keepdim=True))
    return s, v / vn


def pocket_encoder(p, node_s, node_v, edge_index, edge_s, edge_v, seq, prefix="pro_encoder"):
    """GVP_embedding.forward (GVP_Block.py:53-67); messages flow j = edge_index[0] -> i = edge_index[1], mean over the
    incoming edges of i (nodes without incoming edges get zeros), PyG `MessagePassing(aggr='mean')`."""
    s = torch.cat([node_s, p[prefix + ".W_s.weight"][seq]], -1)
    s, v = _gvp_ln(p, prefix + ".W_v.0", s, node_v)
    s, v = _gvp(p, prefix + ".W_v.1", s, v, 16, False, False)
    es, ev = _gvp_ln(p, prefix + ".W_e.0", edge_s, edge_v)
    es, ev = _gvp(p, prefix + ".W_e.1", es, ev, 1, False, False)
    src, dst = edge_index
    n = s.shape[0]
    cnt = _scatter_add(torch.ones(dst.shape[0], 1), dst, n).clamp(min=1)
    for l in range(GVP_LAYERS):
        k = f"{prefix}.layers.{l}"
        ms = torch.cat([s[src], es, s[dst]], -1)
        mv = torch.cat([v[src], ev, v[dst]], -2)
        ms, mv = _gvp(p, k + ".conv.message_func.0", ms, mv, 16, True, True)
        ms, mv = _gvp(p, k + ".conv.message_func.1", ms, mv, 16, True, True)
        ms, mv = _gvp(p, k + ".conv.message_func.2", ms, mv, 16, False, False)
        ds = _scatter_add(ms, dst, n) / cnt
        dv = _scatter_add(mv, dst, n) / cnt.unsqueeze(-1)
        s, v = _gvp_ln(p, k + ".norm.0", s + ds, v + dv)
        fs, fv = _gvp(p, k + ".ff_func.0", s, v, 32, True, True)
        fs, fv = _gvp(p, k + ".ff_func.1", fs, fv, 16, False, False)
        s, v = _gvp_ln(p, k + ".norm.1", s + fs, v + fv)
    s, v = _gvp_ln(p, prefix + ".W_out.0", s, v)
    s, _ = _gvp(p, prefix + ".W_out.1", s, v, 0, True, True)
    return s


# ------------------------------------------------------------------------------------------------ mixture density head
def pair_distance(lig_pos, pro_xyz_full):
    """MDN_Block.compute_euclidean_distances_matrix for ONE graph: min over the 14 atom slots of
    sqrt(|x|^2 + |y|^2 - 2 x.y) in float64, NaN (tiny negative arguments) -> 10000.  Unused slots are NOT masked."""
    X, Y = lig_pos.double(), pro_xyz_full.reshape(-1, 3).double()
    d2 = -2 * X @ Y.T + (Y ** 2).sum(-1)[None, :] + (X ** 2).sum(-1)[:, None]
    d = torch.nan_to_num((d2 ** 0.5).view(X.shape[0], -1, 14), 10000)
    return d.min(-1)[0]


def mdn_score(p, lig_s, lig_pos, lig_batch, pro_s, pro_xyz_full, pro_batch, dist_threshold=5.0, prefix="mdn_layer"):
    """KarmaDock.scoring: per-graph sum over (ligand atom, residue) pairs of the mixture density at their distance."""
    B = int(lig_batch.max()) + 1
    out = torch.zeros(B)
    for b in range(B):
        hl, ht = lig_s[lig_batch == b], pro_s[pro_batch == b]
        C = torch.cat([hl[:, None, :].expand(-1, ht.shape[0], -1), ht[None].expand(hl.shape[0], -1, -1)], -1).reshape(-1, 2 * hl.shape[1])
        C = F.elu(_bn(p, prefix + ".MLP.1", _lin(p, prefix + ".MLP.0", C)))
        pi = F.softmax(_lin(p, prefix + ".z_pi", C), -1)
        sigma = F.elu(_lin(p, prefix + ".z_sigma", C)) + 1.1
        mu = F.elu(_lin(p, prefix + ".z_mu", C)) + 1
        dist = pair_distance(lig_pos[lig_batch == b], pro_xyz_full[pro_batch == b]).reshape(-1, 1)
        # Normal(mu, sigma).log_prob(y) + log(pi); mu / sigma float32, y float64 -> float64 arithmetic (torch promotion)
        logp = -((dist - mu) ** 2) / (2 * sigma ** 2) - torch.log(sigma) - math.log(math.sqrt(2 * math.pi))
        prob = (logp + torch.log(pi)).exp().sum(1)
        prob[dist[:, 0] > dist_threshold] = 0.0
        out[b] = prob.sum().float()
    return out


def forward(p, d):
    """``d``: dict of the tensors KarmaDock.forward reads from its HeteroData batch (covalent ligand edges already selected):
    lig_node_s [NL,89], lig_edge_s [El,20], lig_edge_index [2,El], lig_pos [NL,3], lig_batch [NL], pro_node_s [NR,9],
    pro_node_v [NR,3,3], pro_edge_index [2,Ep], pro_edge_s [Ep,21], pro_edge_v [Ep,1,3], pro_seq [NR], pro_xyz_full
    [NR,14,3], pro_batch [NR].  Returns (score [B], lig_s [NL,128], pro_s [NR,128])."""
    lig_s = ligand_encoder(p, d["lig_node_s"].float(), d["lig_edge_s"].float(), d["lig_edge_index"])
    pro_s = pocket_encoder(p, d["pro_node_s"], d["pro_node_v"], d["pro_edge_index"], d["pro_edge_s"], d["pro_edge_v"], d["pro_seq"])
    score = mdn_score(p, lig_s, d["lig_pos"], d["lig_batch"], pro_s, d["pro_xyz_full"], d["pro_batch"])
    return score, lig_s, pro_s


def param_shapes():
    """name -> shape of every tensor the scorer's forward reads (a subset of KarmaDock's state_dict)."""
    S = {}

    def lin(k, o, i, bias=True):
        S[k + ".weight"] = (o, i)
        if bias:
            S[k + ".bias"] = (o,)

    def bn(k, n):
        for s in ("weight", "bias", "running_mean", "running_var"):
            S[f"{k}.{s}"] = (n,)

    def gvp(k, si, vi, so, vo):
        h = max(vi, vo)
        lin(k + ".wh", h, vi, False)
        lin(k + ".ws", so, h + si)
        if vo:
            lin(k + ".wv", vo, h, False)

    lin("lig_encoder.node_encoder", 128, 89)
    lin("lig_encoder.edge_encoder", 128, 20)
    for l in range(GT_LAYERS):
        k = f"lig_encoder.gt_block.{l}"
        final = l == GT_LAYERS - 1
        for w in ("node", "edge"):
            bn(f"{k}.batch_norm1_{w}_feats", 128)
        for w in ("Q", "K", "V", "edge_feats_projection"):
            lin(f"{k}.mha_module.{w}", 128, 128, False)
        lin(k + ".O_node_feats", 128, 128)
        lin(k + ".node_feats_MLP.0", 256, 128, False)
        lin(k + ".node_feats_MLP.3", 128, 256, False)
        bn(k + ".batch_norm2_node_feats", 128)
        if not final:
            lin(k + ".O_edge_feats", 128, 128)
            bn(k + ".batch_norm2_edge_feats", 128)
            lin(k + ".edge_feats_MLP.0", 256, 128, False)
            lin(k + ".edge_feats_MLP.3", 128, 256, False)
    S["pro_encoder.W_s.weight"] = (31, 31)

    def ln(k, n):
        S[k + ".scalar_norm.weight"] = (n,)
        S[k + ".scalar_norm.bias"] = (n,)

    ln("pro_encoder.W_v.0", 40); gvp("pro_encoder.W_v.1", 40, 3, 128, 16)
    ln("pro_encoder.W_e.0", 21); gvp("pro_encoder.W_e.1", 21, 1, 32, 1)
    for l in range(GVP_LAYERS):
        k = f"pro_encoder.layers.{l}"
        gvp(k + ".conv.message_func.0", 288, 33, 128, 16)
        gvp(k + ".conv.message_func.1", 128, 16, 128, 16)
        gvp(k + ".conv.message_func.2", 128, 16, 128, 16)
        ln(k + ".norm.0", 128); ln(k + ".norm.1", 128)
        gvp(k + ".ff_func.0", 128, 16, 512, 32)
        gvp(k + ".ff_func.1", 512, 32, 128, 16)
    ln("pro_encoder.W_out.0", 128); gvp("pro_encoder.W_out.1", 128, 16, 128, 0)
    lin("mdn_layer.MLP.0", 128, 256); bn("mdn_layer.MLP.1", 128)
    for w in ("z_pi", "z_sigma", "z_mu"):
        lin("mdn_layer." + w, 10, 128)
    return S


def init_params(seed=0):
    """Seeded random weights (the scorer's checkpoint is absent offline): uniform +-1/sqrt(fan_in) like torch's default
    Linear init, BatchNorm statistics / affine parameters moved off their identity values so that they are exercised."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for k, shp in param_shapes().items():
        if k.endswith("running_var"):
            P[k] = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("running_mean"):
            P[k] = 0.2 * torch.randn(shp, generator=g)
        elif "norm" in k.split(".")[-2] and k.endswith("weight") or (".MLP.1." in k and k.endswith("weight")) or "batch_norm" in k and k.endswith("weight"):
            P[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            P[k] = 0.1 * torch.randn(shp, generator=g)
        elif k == "pro_encoder.W_s.weight":
            P[k] = torch.randn(shp, generator=g)
        else:
            P[k] = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(shp[1])
    return P
