"""``KarmaDockHIP``: the MDN pose scorer's network forward on the device.

Mirrors ``KarmaDock`` of DiffBindFR/scoring/architecture/KarmaDock_sc.py (constructed in
DiffBindFR/common/engines.py:249, weights loaded with ``strict=False``, :263-268): same parameter names for everything
the scoring forward reads (``lig_encoder.*``, ``pro_encoder.*``, ``mdn_layer.*``), ``forward(data) -> mdn_score [B]`` with
``data`` the featurised HeteroData batch (or a plain dict of the same tensors).  The arithmetic runs in libdbfr.so
(csrc/mdn.hip) through ``dbfr_mdn_forward``; there is no CPU path.  The RDKit / openfold featurisation that builds
``data`` stays with the reference.
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from . import lib as L
from .registry import ENERGY

GT_LAYERS, GVP_LAYERS = 6, 3


def param_shapes():
    """name -> shape of every tensor of KarmaDock's state_dict that its scoring forward reads."""
    S = {}

    def lin(k, o, i, bias=True):
        S[k + ".weight"] = (o, i)
        if bias:
            S[k + ".bias"] = (o,)

    def bn(k, n):
        for s in ("weight", "bias", "running_mean", "running_var"):
            S[f"{k}.{s}"] = (n,)

    def ln(k, n):
        S[k + ".scalar_norm.weight"] = (n,)
        S[k + ".scalar_norm.bias"] = (n,)

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
        for w in ("node", "edge"):
            bn(f"{k}.batch_norm1_{w}_feats", 128)
        for w in ("Q", "K", "V", "edge_feats_projection"):
            lin(f"{k}.mha_module.{w}", 128, 128, False)
        lin(k + ".O_node_feats", 128, 128)
        lin(k + ".node_feats_MLP.0", 256, 128, False)
        lin(k + ".node_feats_MLP.3", 128, 256, False)
        bn(k + ".batch_norm2_node_feats", 128)
        if l < GT_LAYERS - 1:
            lin(k + ".O_edge_feats", 128, 128)
            bn(k + ".batch_norm2_edge_feats", 128)
            lin(k + ".edge_feats_MLP.0", 256, 128, False)
            lin(k + ".edge_feats_MLP.3", 128, 256, False)
    S["pro_encoder.W_s.weight"] = (31, 31)
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


class _Store(nn.Module):
    """Nested parameter container: attribute path == state_dict key."""


def _put(root, name, tensor, buffer):
    parts = name.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Store())
        m = getattr(m, p)
    if buffer:
        m.register_buffer(parts[-1], tensor)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def batch_from_hetero(data):
    """The tensors ``KarmaDock.forward`` reads from its HeteroData batch (KarmaDock_sc.py:58-96) as a flat dict."""
    lig, pro = data["ligand"], data["protein"]
    l2l, p2p = data[("ligand", "l2l", "ligand")], data[("protein", "p2p", "protein")]
    g = lambda s, k: s[k] if isinstance(s, dict) else getattr(s, k)
    cov = g(lig, "cov_edge_mask")
    return dict(lig_node_s=g(lig, "node_s"), lig_edge_s=g(l2l, "edge_s")[cov], lig_edge_index=g(l2l, "edge_index")[:, cov],
                lig_pos=g(lig, "xyz"), lig_batch=g(lig, "batch"), pro_node_s=g(pro, "node_s"), pro_node_v=g(pro, "node_v"),
                pro_edge_index=g(p2p, "edge_index"), pro_edge_s=g(p2p, "edge_s"), pro_edge_v=g(p2p, "edge_v"),
                pro_seq=g(pro, "seq"), pro_xyz_full=g(pro, "xyz_full"), pro_batch=g(pro, "batch"))


@torch.no_grad()
def pocket_features(aatype, atom14_pos, res_ptr=None, topk=30):
    """The pocket half of the scorer's input on the device (``dbfr_mdn_pocket_features``): replaces ``get_protein_feature``
    (DiffBindFR/scoring/dataset/protein_feature.py:137-216) behind its PDB parser for any number of pockets / poses at once.
    aatype [N] (< 20), atom14_pos [N,14,3] device tensor with absent atoms at the origin (``dbfr_sample``'s atom14 output plus the
    pocket centre), res_ptr [P+1] (None = one pocket).  Returns the ``pro_*`` entries of the flat batch dict ``score`` takes."""
    lib = L.load()
    dev = atom14_pos.device
    if dev.type != "cuda":
        raise L.DbfrError("pocket_features needs a ROCm device (no CPU path)")
    n = int(atom14_pos.shape[0])
    rp = torch.tensor([0, n]) if res_ptr is None else torch.as_tensor(res_ptr).cpu().long()
    cnt = rp[1:] - rp[:-1]
    if int(cnt.max()) > 1024 or int(cnt.min()) < 2:
        raise L.DbfrError("pockets of 2 .. 1024 residues")
    per = cnt * torch.minimum(torch.full_like(cnt, int(topk)), cnt - 1)
    ep = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(per, 0)])
    E, P = int(ep[-1]), int(cnt.numel())
    i32 = lambda x: torch.as_tensor(x).to(device=dev, dtype=torch.int32).contiguous()
    aa, x = i32(aatype), atom14_pos.to(torch.float32).contiguous()
    rpd, epd = i32(rp), i32(ep)
    out = dict(pro_node_s=torch.empty(n, 9, device=dev), pro_node_v=torch.empty(n, 3, 3, device=dev),
               pro_edge_s=torch.empty(E, 21, device=dev), pro_edge_v=torch.empty(E, 1, 3, device=dev))
    src, dst = torch.empty(E, dtype=torch.int32, device=dev), torch.empty(E, dtype=torch.int32, device=dev)
    in_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        L.check(lib.dbfr_mdn_pocket_features(P, n, p(rpd), p(epd), p(aa), p(x), int(topk), p(out["pro_node_s"]), p(out["pro_node_v"]),
                                             p(src), p(dst), p(in_ptr), p(out["pro_edge_s"]), p(out["pro_edge_v"]),
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    out.update(pro_edge_index=torch.stack([src, dst]).long(), pro_seq=aa.long(), pro_xyz_full=x,
               pro_batch=torch.repeat_interleave(torch.arange(P, device=dev), cnt.to(dev)))
    return out


@ENERGY.register_module(name=["KarmaDockHIP"])
class KarmaDockHIP(nn.Module):
    def __init__(self, cfg=None, **kwargs):
        super().__init__()
        self.cfg = cfg
        for k, shp in param_shapes().items():
            buf = k.endswith(("running_mean", "running_var"))
            _put(self, k, torch.ones(shp) if k.endswith(("running_var", "norm.weight")) else torch.zeros(shp), buf)
        self._handles = {}
        self._ws = {}

    def load_state_dict(self, state_dict, strict=False, **kw):
        """The reference loads the scorer with strict=False (engines.py:263-268): its checkpoint carries the pose-prediction
        modules (egnn_layers, gates, torsion head) that the scoring forward never calls."""
        mine = set(self.state_dict().keys())
        return super().load_state_dict({k: v for k, v in state_dict.items() if k in mine or strict}, strict=strict, **kw)

    def release(self):
        for _, h in self._handles.values():
            L.load().dbfr_mdn_model_destroy(h)
        self._handles = {}

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def handle(self, device):
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        fp = tuple((v.data_ptr(), v._version) for v in self.state_dict(keep_vars=True).values())
        cur = self._handles.get(idx)
        if cur is not None and cur[0] == fp:
            return cur[1]
        lib = L.load()
        if cur is not None:
            lib.dbfr_mdn_model_destroy(cur[1])
            del self._handles[idx]
        sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in self.state_dict().items()}
        arr = (L.Tensor * len(sd))()
        for i, (k, v) in enumerate(sd.items()):
            arr[i].name, arr[i].data, arr[i].numel = k.encode(), C.c_void_p(v.data_ptr()), v.numel()
        h = C.c_void_p()
        with torch.cuda.device(idx):
            L.check(lib.dbfr_mdn_model_create(arr, len(sd), C.byref(h)))
        self._handles[idx] = (fp, h)
        return h

    @torch.no_grad()
    def score(self, d, lig_s=None, return_embeddings=False, dist_threshold=5.0):
        """``d``: flat dict (``batch_from_hetero``) of DEVICE tensors.  Returns mdn_score [B] (and the embeddings)."""
        lib = L.load()
        dev = d["lig_pos"].device
        if dev.type != "cuda":
            raise L.DbfrError("KarmaDockHIP needs a ROCm device (no CPU path)")
        i32 = lambda x: x.to(device=dev, dtype=torch.int32).contiguous()
        f32 = lambda x: x.to(device=dev, dtype=torch.float32).contiguous()
        lb, pb = d["lig_batch"].to(dev).long(), d["pro_batch"].to(dev).long()
        B = int(lb.max().item()) + 1
        ptr = lambda b: torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(torch.bincount(b, minlength=B), 0)])
        assert bool((lb[1:] >= lb[:-1]).all()) and bool((pb[1:] >= pb[:-1]).all()), "graphs must be contiguous node ranges"
        NL, NR = int(lb.numel()), int(pb.numel())
        lei = d["lig_edge_index"].to(dev).long()
        EL = int(lei.shape[1])
        order = torch.argsort(lei[1], stable=True)                   # edge ids grouped by col, ascending inside a group
        lin_ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(torch.bincount(lei[1], minlength=NL), 0)])
        pei = d["pro_edge_index"].to(dev).long()
        EP = int(pei.shape[1])
        perm = torch.argsort(pei[1], stable=True)                    # pocket edges grouped by target (no-op for knn_graph output)
        pei = pei[:, perm]
        pin_ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(torch.bincount(pei[1], minlength=NR), 0)])
        one = lambda *s: torch.zeros(*s, device=dev)
        T = dict(lig_ptr=i32(ptr(lb)), lig_node_s=f32(d["lig_node_s"]), lig_edge_s=f32(d["lig_edge_s"]) if EL else one(1, 20),
                 lig_edge_src=i32(lei[0]) if EL else i32(torch.zeros(1)), lig_edge_dst=i32(lei[1]) if EL else i32(torch.zeros(1)),
                 lig_in_ptr=i32(lin_ptr), lig_in_edge=i32(order) if EL else i32(torch.zeros(1)), lig_pos=f32(d["lig_pos"]),
                 lig_s_in=None if lig_s is None else f32(lig_s), res_ptr=i32(ptr(pb)), pro_node_s=f32(d["pro_node_s"]),
                 pro_node_v=f32(d["pro_node_v"]), pro_edge_src=i32(pei[0]) if EP else i32(torch.zeros(1)),
                 pro_edge_dst=i32(pei[1]) if EP else i32(torch.zeros(1)), pro_in_ptr=i32(pin_ptr),
                 pro_edge_s=f32(d["pro_edge_s"].to(dev)[perm]) if EP else one(1, 21),
                 pro_edge_v=f32(d["pro_edge_v"].to(dev)[perm]) if EP else one(1, 3), pro_seq=i32(d["pro_seq"]),
                 pro_xyz_full=f32(d["pro_xyz_full"]))
        b = L.MdnBatch(B=B, NL=NL, EL=EL, NR=NR, EP=EP, dist_threshold=float(dist_threshold))
        for k in L._MDN_PTRS:
            setattr(b, k, None if T[k] is None else C.c_void_p(T[k].data_ptr()))
        nbytes = C.c_size_t()
        L.check(lib.dbfr_mdn_workspace_bytes(C.byref(b), C.byref(nbytes)))
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        ws = self._ws.get(idx)
        if ws is None or ws.numel() < nbytes.value:
            ws = self._ws[idx] = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        score = torch.empty(B, device=dev)
        ls = torch.empty(NL, 128, device=dev) if return_embeddings else None
        ps = torch.empty(NR, 128, device=dev) if return_embeddings else None
        p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            L.check(lib.dbfr_mdn_forward(self.handle(dev), C.byref(b), p(score), p(ls), p(ps), C.c_void_p(ws.data_ptr()), ws.numel(), stream))
        self._keep = T                      # inputs stay alive until the stream has consumed them
        return (score, ls, ps) if return_embeddings else score

    @torch.no_grad()
    def score_poses(self, lig, aatype, atom14_poses, lig_poses, topk=30, reuse_ligand_embeddings=True):
        """Scores P poses of ONE complex where the sampler left them: ``atom14_poses`` [P,N_r,14,3] and ``lig_poses`` [P,N_l,3]
        device tensors in absolute coordinates (``dbfr_sample`` output + the pocket centre, absent atom14 slots zero), ``aatype``
        [N_r]; ``lig``: the ligand's pose-independent features from the reference's featuriser (``lig_node_s`` [N_l,89],
        ``lig_edge_s`` [E,20], ``lig_edge_index`` [2,E], covalent edges).  The pocket features of all poses come from one
        ``dbfr_mdn_pocket_features`` launch, the ligand encoder runs once.  Returns mdn_score [P]."""
        dev = atom14_poses.device
        P, n_r = int(atom14_poses.shape[0]), int(atom14_poses.shape[1])
        n_l = int(lig_poses.shape[1])
        f = pocket_features(torch.as_tensor(aatype).to(dev).repeat(P), atom14_poses.reshape(P * n_r, 14, 3),
                            res_ptr=torch.arange(P + 1) * n_r, topk=topk)
        ei = torch.as_tensor(lig["lig_edge_index"]).to(dev).long()
        off = (torch.arange(P, device=dev) * n_l).repeat_interleave(ei.shape[1])
        d = dict(f, lig_node_s=torch.as_tensor(lig["lig_node_s"]).to(dev).float().repeat(P, 1),
                 lig_edge_s=torch.as_tensor(lig["lig_edge_s"]).to(dev).float().repeat(P, 1),
                 lig_edge_index=ei.repeat(1, P) + off, lig_pos=lig_poses.reshape(P * n_l, 3),
                 lig_batch=torch.arange(P, device=dev).repeat_interleave(n_l))
        lig_s = None
        if reuse_ligand_embeddings and P > 1:
            one = {k: (v[:n_l] if k in ("lig_node_s", "lig_pos", "lig_batch") else v) for k, v in d.items()}
            one.update(lig_edge_s=d["lig_edge_s"][:ei.shape[1]], lig_edge_index=ei)
            m0 = f["pro_batch"] == 0
            e0 = f["pro_batch"][f["pro_edge_index"][1]] == 0
            one.update(pro_node_s=f["pro_node_s"][m0], pro_node_v=f["pro_node_v"][m0], pro_seq=f["pro_seq"][m0], pro_xyz_full=f["pro_xyz_full"][m0],
                       pro_batch=f["pro_batch"][m0], pro_edge_index=f["pro_edge_index"][:, e0], pro_edge_s=f["pro_edge_s"][e0], pro_edge_v=f["pro_edge_v"][e0])
            _, ls, _ = self.score(one, return_embeddings=True)
            lig_s = ls.repeat(P, 1)
        return self.score(d, lig_s=lig_s)

    @torch.no_grad()
    def score_complexes(self, items, topk=30):
        """``score_poses`` for several complexes in ONE pocket-feature launch and ONE network forward.
        ``items``: list of (lig, aatype, atom14_poses [P_c,N_r,14,3], lig_poses [P_c,N_l,3]).  Returns list of score tensors [P_c]."""
        dev = items[0][2].device
        aa, x14, rp = [], [], [0]
        L_ = {k: [] for k in ("lig_node_s", "lig_edge_s", "lig_edge_index", "lig_pos", "lig_batch")}
        n_graph = lo = 0
        for lig, aatype, a14, lp in items:
            P, n_r, n_l = int(a14.shape[0]), int(a14.shape[1]), int(lp.shape[1])
            aa.append(torch.as_tensor(aatype).to(dev).repeat(P))
            x14.append(a14.reshape(P * n_r, 14, 3))
            rp += [rp[-1] + n_r * (i + 1) for i in range(P)]
            ei = torch.as_tensor(lig["lig_edge_index"]).to(dev).long()
            L_["lig_node_s"].append(torch.as_tensor(lig["lig_node_s"]).to(dev).float().repeat(P, 1))
            L_["lig_edge_s"].append(torch.as_tensor(lig["lig_edge_s"]).to(dev).float().repeat(P, 1))
            L_["lig_edge_index"].append(ei.repeat(1, P) + lo + (torch.arange(P, device=dev) * n_l).repeat_interleave(ei.shape[1]))
            L_["lig_pos"].append(lp.reshape(P * n_l, 3))
            L_["lig_batch"].append(n_graph + torch.arange(P, device=dev).repeat_interleave(n_l))
            lo += P * n_l
            n_graph += P
        rp = [0] + list(np.cumsum([int(it[2].shape[1]) for it in items for _ in range(int(it[2].shape[0]))]))
        f = pocket_features(torch.cat(aa), torch.cat(x14), res_ptr=rp, topk=topk)
        d = dict(f, **{k: torch.cat(v, 1 if k == "lig_edge_index" else 0) for k, v in L_.items()})
        s = self.score(d)
        out, g = [], 0
        for it in items:
            P = int(it[2].shape[0])
            out.append(s[g:g + P])
            g += P
        return out

    def forward(self, data):
        """KarmaDock.forward (KarmaDock_sc.py:58-70)."""
        return self.score(batch_from_hetero(data) if not isinstance(data, dict) or "ligand" in data else data)


# ------------------------------------------------------------------------------------------------ the Scorer entry
def scorer_state_dict(ckpt):
    """The scorer's parameters out of whatever the reference's loaders read (DiffBindFR/scoring/utils/early_stop.py:27-38):
      * ``{'model': {...}}`` -- ``Early_stopper.load_model(mine=True)``, the route ``Scorer`` takes with ``mdn_paper.pt``
        (engines.py:262-268): every key loses its first SIX characters (the saved wrapper's prefix) and gets ``module.`` for the
        DataParallel model the reference loads into -- here the bare name remains;
      * ``{'model_state_dict': {...}}`` -- ``mine=False`` / ``save_model``;
      * a flat state_dict.
    A ``module.`` prefix (DataParallel) is dropped in every case."""
    if not isinstance(ckpt, dict):
        raise TypeError(f"scorer checkpoint: expected a dict, got {type(ckpt).__name__}")
    if "model" in ckpt and isinstance(ckpt["model"], dict):
        sd = {k[6:]: v for k, v in ckpt["model"].items()}
    elif "model_state_dict" in ckpt and isinstance(ckpt["model_state_dict"], dict):
        sd = ckpt["model_state_dict"]
    else:
        sd = ckpt
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_scorer_weights(model, ckpt, min_fraction=0.9):
    """``strict=False`` like the reference (its checkpoint carries pose-prediction modules the scoring forward never calls) -- but a load
    that matches (almost) none of the model's parameters is an error here, not a silent run on the initial parameters: raises unless
    at least ``min_fraction`` of ``model``'s parameters and buffers were found with their shapes.  Returns the matched key count."""
    sd = scorer_state_dict(ckpt)
    mine = model.state_dict()
    hit = [k for k in mine if k in sd and tuple(sd[k].shape) == tuple(mine[k].shape)]
    if len(hit) < min_fraction * len(mine):
        bad = [k for k in mine if k in sd and tuple(sd[k].shape) != tuple(mine[k].shape)]
        raise L.DbfrError(f"scorer checkpoint: {len(hit)} of {len(mine)} parameters found (first missing: "
                          f"{[k for k in mine if k not in sd][:3]}, shape mismatches: {bad[:3]}, checkpoint keys look like {list(sd)[:3]}); "
                          f"refusing to score with the initial parameters")
    model.load_state_dict(sd, strict=False)
    return len(hit)


def collate_flat(items):
    """Concatenate flat per-pair dicts (``batch_from_hetero`` layout, one graph each or already batched) into one batch --
    what the reference's PyG DataLoader does with the HeteroData samples (engines.py:270-277): node tensors stacked, edge
    indices shifted by the node offsets, ``*_batch`` renumbered.  Every item must carry the same keys (``lig_batch`` /
    ``pro_batch`` may be absent: one graph)."""
    keys = set(items[0])
    for i, it in enumerate(items):
        if set(it) != keys:
            raise KeyError(f"collate_flat: item {i} has keys {sorted(set(it) ^ keys)} that the others lack (or lacks theirs)")
    out = {k: [] for k in items[0]}
    lo = po = g = 0
    for it in items:
        nl, nr = int(it["lig_pos"].shape[0]), int(it["pro_node_s"].shape[0])
        lb = torch.as_tensor(it.get("lig_batch", torch.zeros(nl, dtype=torch.long))).long()
        pb = torch.as_tensor(it.get("pro_batch", torch.zeros(nr, dtype=torch.long))).long()
        for k, v in it.items():
            v = torch.as_tensor(v)
            if k == "lig_edge_index":
                v = v.long() + lo
            elif k == "pro_edge_index":
                v = v.long() + po
            elif k == "lig_batch":
                v = lb + g
            elif k == "pro_batch":
                v = pb + g
            out[k].append(v)
        if "lig_batch" not in it:
            out.setdefault("lig_batch", []).append(lb + g)
        if "pro_batch" not in it:
            out.setdefault("pro_batch", []).append(pb + g)
        lo += nl; po += nr
        g += int(lb.max()) + 1 if nl else 1
    return {k: torch.cat(v, 1 if k.endswith("edge_index") else 0) for k, v in out.items() if v}


def Scorer(test_dataset, model_weight=None, output_path="mdn_score.csv", batch_size=16, device_id=0, logger=None, model=None):
    """Drop-in shape of DiffBindFR/common/engines.py:230-302: score every (pocket, ligand pose) sample of ``test_dataset`` with
    the MDN network and write ``test_dataset.pair_frame['mdn_score']`` to ``output_path``.

    ``test_dataset``: indexable, ``len()``; item i is the featurised sample of pair i -- the reference's HeteroData (anything
    ``batch_from_hetero`` reads) or the flat dict -- and ``pair_frame`` a pandas frame with one row per item (optional).
    ``model_weight``: a checkpoint dict, a path ``torch.load`` reads, or None (keep ``model``'s weights) -- in any of the layouts the
    reference's ``Early_stopper.load_model`` reads (``scorer_state_dict``: ``mdn_paper.pt`` is ``{'model': {'<6-char prefix><name>': ...}}``);
    unknown keys are ignored (``strict=False`` like engines.py:263-268), but a checkpoint that matches less than 90 % of the model's
    parameters raises (``load_scorer_weights``) instead of scoring with the initial parameters.
    Batches of ``batch_size`` samples are collated here (``collate_flat``) and go through ONE ``dbfr_mdn_forward`` each.  A sample that
    is ``None`` (failed featurisation) is left out of its batch like the reference's PassNoneDataLoader does
    (scoring/dataset/dataloader.py:16-17) and its row gets NaN -- the reference's frame assignment would raise on the length mismatch.
    Returns the list of scores (the reference returns None after writing the file)."""
    dev = torch.device(f"cuda:{device_id}" if not isinstance(device_id, torch.device) else device_id)
    model = model or KarmaDockHIP()
    if logger is not None:
        logger.info("Load scoring model...")
    if model_weight is not None:
        ckpt = model_weight if isinstance(model_weight, dict) else torch.load(str(model_weight), map_location="cpu")
        n_hit = load_scorer_weights(model, ckpt)
        if logger is not None:
            logger.info(f"scoring model: {n_hit} tensors loaded")
    scores = []
    n = len(test_dataset)
    for i0 in range(0, n, batch_size):
        items = [test_dataset[i] for i in range(i0, min(n, i0 + batch_size))]
        keep = [k for k, it in enumerate(items) if it is not None]
        part = [float("nan")] * len(items)
        if keep:
            flat = [items[k] if isinstance(items[k], dict) and "ligand" not in items[k] else batch_from_hetero(items[k]) for k in keep]
            d = {k: v.to(dev) for k, v in collate_flat(flat).items()}
            got = model.score(d).cpu().numpy().tolist()
            assert len(got) == len(keep), "one sample = one graph"
            for k, v in zip(keep, got):
                part[k] = v
        scores.extend(part)
    frame = getattr(test_dataset, "pair_frame", None)
    if frame is not None:
        frame["mdn_score"] = scores
        if output_path is not None:
            frame.to_csv(str(output_path), index=False)
    if logger is not None:
        logger.info("Model Scoring is Done!")
    return scores
