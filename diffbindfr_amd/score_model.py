"""``TensorProductModelHIP``: drop-in for INTERACTION['TensorProductModel'].

Level-1 boundary of SURVEY.md section 8(b): same constructor (``cfg`` = the ConfigDict of
DiffBindFR/configs/diffbindfr_ts.py:107-142), same ``forward(data) -> (tr, rot, tor,
sc_tor)``, same ``state_dict`` keys as druglib/models/Docking/interaction/tpscore.py:202-573, and
``tp`` / ``final_tp_tor`` children that absorb the e3nn buffers of a real checkpoint, so
the reference's own loader (druglib/core/runner/checkpoint.py:32-100: a recursive
``_load_from_state_dict`` walk, ``strict=True`` from DiffBindFR/app/predict.py:118-125)
accepts it -- pinned by tests/test_host.py::test_reference_loader_fixture; but
the arithmetic runs in libdbfr.so (hand-written HIP for gfx950) through the C ABI of
include/dbfr.h.  There is no CPU path: without the library / a GPU it raises.
"""
import ctypes as C

import torch
from torch import nn

from . import lib as L
from .packing import PackedBatch, _get, _has
from .registry import INTERACTION


def cfg_get(cfg, path, default=None):
    cur = cfg
    for k in path.split("."):
        if cur is None:
            return default
        if isinstance(cur, dict):
            cur = cur.get(k)
        else:
            cur = getattr(cur, k, None)
    return default if cur is None else cur


class _SimpleLinear(nn.Module):
    """Parameter container with the reference's key layout (tpscore.py:109-141)."""

    def __init__(self, i, o, h=None, bias=True, act="relu"):
        super().__init__()
        h = h or o
        self.lin = nn.Sequential(nn.Linear(i, h, bias=bias), nn.ReLU() if act == "relu" else nn.Tanh(),
                                 nn.Dropout(0.0), nn.Linear(h, o, bias=bias))


class _LayerNormParams(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        n = sum(m for m, l, p in blocks)
        n0e = sum(m for m, l, p in blocks if l == 0 and p == 1)
        ms = torch.cat([torch.ones(m) if (l == 0 and p == 1) else torch.zeros(m) for m, l, p in blocks])
        self.mean_shift = nn.Parameter(ms.view(1, n, 1))
        self.affine_weight = nn.Parameter(torch.ones(n))
        self.affine_bias = nn.Parameter(torch.zeros(n0e))


class _KeySink(nn.Module):
    """Stands where the reference holds an e3nn module (``TensorProductConvLayer.tp`` tpscore.py:163,
    ``final_tp_tor`` :373).  With ``shared_weights=False`` those own no parameters, only buffers whose names depend
    on the e3nn build (``weight`` (empty), ``output_mask``, Wigner-3j constants of the generated code).  The
    reference's loader walks ``_load_from_state_dict`` module by module: this child takes every key under its
    prefix so that ``strict=True`` passes, and remembers them (``absorbed``)."""

    def __init__(self):
        super().__init__()
        self.absorbed = []

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        self.absorbed = [k for k in state_dict if k.startswith(prefix)]


class _Conv(nn.Module):
    def __init__(self, nef, W, out_blocks):
        super().__init__()
        self.tp = _KeySink()
        self.fc = _SimpleLinear(nef, W, nef)
        self.batch_norm = _LayerNormParams(out_blocks)


class _Smearing(nn.Module):
    def __init__(self, stop, n):
        super().__init__()
        off = torch.linspace(0.0, stop, n)
        self.register_buffer("coeff", -0.5 / (off[1] - off[0]) ** 2)
        self.register_buffer("offset", off)


class _AtomEncoder(nn.Module):
    def __init__(self, ns, dims, scalar_dim):
        super().__init__()
        self.atom_emb_list = nn.ModuleList([nn.Embedding(d, ns) for d in dims])
        self.scalar_lin = nn.Linear(scalar_dim + ns, ns, bias=False)


def conv_paths(kind):
    """(weight_numel, path table) as the library derives them (dbfr_conv_paths)."""
    lib = L.load()
    tab = (C.c_int32 * (16 * 10))()
    wn = C.c_int32()
    n = lib.dbfr_conv_paths(kind, tab, 16, C.byref(wn))
    if n < 0:
        L.check(n)
    return wn.value, [list(tab[10 * i:10 * i + 10]) for i in range(n)]


GEMM_MODES = {"f32": 0, "split_f16": 3, "reduce_first": 4}     # include/dbfr.h: DBFR_GEMM_*


@INTERACTION.register_module(name=["TensorProductModelHIP"])
class TensorProductModelHIP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        g = lambda k, d=None: cfg_get(cfg, k, d)
        assert g("task", "struct_gen") == "struct_gen", "only the score-matching task is on this path"
        assert not g("use_second_order_repr", False), "use_second_order_repr=True is not supported"
        self.no_sc_torsion = bool(g("no_sc_torsion", False))
        # which matrix instruction carries the radial MLP's big GEMM: None = the library's default (or $DBFR_GEMM),
        # "f32" = v_mfma_f32_16x16x4_f32,
        # "split_f16" = two fp16 pieces / three products on v_mfma_f32_16x16x32_f16, "reduce_first" (the default) = the same arithmetic with the
        # scalar-output rows reduced over a target's edges before the big GEMM (include/dbfr.h)
        self.gemm = g("gemm", None)
        ns, nv = int(g("ns", 48)), int(g("nv", 12))
        se, de = int(g("sigma_embed_dim", 32)), int(g("distance_embed_dim", 32))
        self.ns = ns
        self.num_conv_layers = int(g("num_conv_layers", 6))
        nl = int(g("features_dim.ligand_atom.node_features", 27))
        ne = int(g("features_dim.ligand_atom.edge_features", 10))
        feat = g("features_dim.protein_atom.feature_list", ((37, 22, 4, 21, 2), 0))
        self.mcfg = L.ModelCfg(
            ns=ns, nv=nv, sh_lmax=int(g("sh_lmax", 2)), num_conv_layers=self.num_conv_layers,
            lig_node_features=nl, lig_edge_features=ne, distance_embed_dim=de, sigma_embed_dim=se,
            emb_scale=float(g("emb_scale", 1000)), lig_cutoff=float(g("lig_cutoff", 5)),
            atom_cutoff=float(g("atom_cutoff", 4)), cross_cutoff=float(g("cross_cutoff", 32)),
            center_max_distance=float(g("center_max_distance", 32)),
            atom_max_neighbors=int(g("atom_max_neighbors", 1000)), lig_max_neighbors=32,
            dynamic_max_cross=int(bool(g("dynamic_max_cross", True))), scale_by_sigma=int(bool(g("scale_by_sigma", True))),
            no_sc_torsion=int(self.no_sc_torsion))
        # ---- parameters, named exactly like the reference module tree
        self.lig_node_embedding = _SimpleLinear(nl + se, ns)
        self.lig_edge_embedding = _SimpleLinear(ne + se + de, ns)
        self.atom_node_embedding = _AtomEncoder(ns, tuple(feat[0]), int(feat[1]) + se)
        self.atom_edge_embedding = _SimpleLinear(se + de, ns)
        self.la_edge_embedding = _SimpleLinear(se + de, ns)
        self.lig_distance_expansion = _Smearing(float(g("lig_cutoff", 5)), de)
        self.atom_distance_expansion = _Smearing(float(g("atom_cutoff", 4)), de)
        self.cross_distance_expansion = _Smearing(float(g("cross_cutoff", 32)), de)
        seq = [[(ns, 0, 1)], [(ns, 0, 1), (nv, 1, -1)], [(ns, 0, 1), (nv, 1, -1), (nv, 1, 1)],
               [(ns, 0, 1), (nv, 1, -1), (nv, 1, 1), (ns, 0, -1)]]
        wn = [conv_paths(k)[0] for k in range(6)]
        fams = ("lig_conv_layers", "atom_conv_layers", "cross_al_conv_layers", "cross_la_conv_layers")
        for f in fams:
            setattr(self, f, nn.ModuleList())
        for l in range(self.num_conv_layers):
            for f in fams:
                getattr(self, f).append(_Conv(3 * ns, wn[min(l, 3)], seq[min(l + 1, 3)]))
        self.center_distance_expansion = _Smearing(float(g("center_max_distance", 32)), de)
        self.center_edge_embedding = _SimpleLinear(de + se, ns)
        self.final_conv = _Conv(2 * ns, wn[4], [(2, 1, -1), (2, 1, 1)])
        self.tr_final_layer = _SimpleLinear(1 + se, 1, ns)
        self.rot_final_layer = _SimpleLinear(1 + se, 1, ns)
        self.final_tp_tor = _KeySink()
        self.tor_edge_embedding = _SimpleLinear(de, ns)
        self.tor_bond_conv = _Conv(3 * ns, wn[5], [(ns, 0, -1), (ns, 0, 1)])
        self.tor_final_layer = _SimpleLinear(2 * ns, 1, ns, bias=False, act="tanh")
        if not self.no_sc_torsion:
            self.sc_edge_embedding = _SimpleLinear(de, ns)
            self.sc_tor_bond_conv = _Conv(3 * ns, wn[5], [(ns, 0, -1), (ns, 0, 1)])
            self.sc_tor_final_layer = _SimpleLinear(2 * ns, 1, ns, bias=False, act="tanh")
        self._handles = {}      # device index -> (fingerprint of the parameters, dbfr_model*)
        self._ws = {}           # device index -> uint8 workspace tensor
        self.limits = L.Limits(24, 64)
        self.auto_grow = True   # on DBFR_ERR_CAPACITY: raise the limits to what the device counted and resume
        self.regrown = 0        # how often that happened (tests / diagnostics)

    def grow_limits(self, pb, ws, stream):
        """After DBFR_ERR_CAPACITY: read what the step needed (``dbfr_capacity_report``), raise ``self.limits`` with 25 %
        head-room (they stay raised for later batches).  Returns the first step that has to be re-run."""
        lib = L.load()
        first, need = C.c_int32(), (C.c_int64 * 8)()
        L.check(lib.dbfr_capacity_report(C.c_void_p(ws.data_ptr()), stream, C.byref(first), need))
        d = pb.dims
        aa = -(-int(need[1] * 1.25) // max(d["NA"], 1)) + 1
        cross = -(-int(max(need[2], need[6]) * 1.25) // max(d["NL"], 1)) - 2 * d["max_nr"] + 1
        new = L.Limits(max(self.limits.aa_avg_neighbors, aa), max(self.limits.cross_avg_neighbors, cross))
        if (new.aa_avg_neighbors, new.cross_avg_neighbors) == (self.limits.aa_avg_neighbors, self.limits.cross_avg_neighbors):
            raise L.DbfrError(f"DBFR_ERR_CAPACITY with nothing to grow (needed {list(need)})")
        self.limits = new
        self.regrown += 1
        return max(int(first.value), 0)

    @property
    def ignored_keys(self):
        """Checkpoint keys the e3nn stand-ins took during the last load."""
        return [k for m in self.modules() if isinstance(m, _KeySink) for k in m.absorbed]

    def release(self):
        for _, h in self._handles.values():
            L.load().dbfr_model_destroy(h)
        self._handles = {}

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def _fingerprint(self):
        # whatever route new weights take (load_state_dict, the reference's per-module loader, .to(), in-place
        # edits), either the storage or the version counter of a tensor changes
        return tuple((v.data_ptr(), v._version) for v in self.state_dict(keep_vars=True).values())

    def handle(self, device=None):
        """The device-resident packed model of the CURRENT parameters on ``device`` (default: the current HIP device).
        Re-packed when a parameter changed since the last call; one handle per device."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        fp = self._fingerprint()
        cur = self._handles.get(idx)
        if cur is not None and cur[0] == fp:
            return cur[1]
        lib = L.load()
        if cur is not None:
            lib.dbfr_model_destroy(cur[1])
            del self._handles[idx]
        sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in self.state_dict().items()}
        arr = (L.Tensor * len(sd))()
        for i, (k, v) in enumerate(sd.items()):
            arr[i].name = k.encode()
            arr[i].data = C.c_void_p(v.data_ptr())
            arr[i].numel = v.numel()
        h = C.c_void_p()
        with torch.cuda.device(idx):            # the library allocates on the current device
            L.check(lib.dbfr_model_create(C.byref(self.mcfg), arr, len(sd), C.byref(h)))
        if self.gemm is not None:
            L.check(lib.dbfr_model_set_gemm(h, GEMM_MODES[self.gemm]))
        log = getattr(self, "_edge_log", None)
        if isinstance(log, dict) and ("cuda", idx) in log:      # the edge-count read-out survives a re-pack
            t = log[("cuda", idx)]
            L.check(lib.dbfr_model_set_edge_log(h, C.c_void_p(t.data_ptr()), t.shape[0], t.shape[2]))
        tl = getattr(self, "_tie_log", None)
        if isinstance(tl, dict) and ("cuda", idx) in tl:        # ... and the near-tie read-out
            t, tol = tl[("cuda", idx)]
            L.check(lib.dbfr_model_set_tie_log(h, C.c_void_p(t.data_ptr()), t.shape[0], t.shape[2], tol))
        self._handles[idx] = (fp, h)
        return h

    def rowscaled_convs(self, device=None):
        """{state_dict prefix: depth} of the convs that hold a run whose rows lie more than 2^17 apart (depth = log2 of the spread found) and
        are therefore packed with one power-of-two factor per ROW, taken off the accumulator rows by the kernel
        (``dbfr_model_rowscaled_convs``).  Empty for seeded weights; a trained checkpoint may name some."""
        buf = C.create_string_buffer(8192)
        n = L.load().dbfr_model_rowscaled_convs(self.handle(device), buf, 8192)
        if n < 0:
            L.check(n)
        return {x.split(":")[0]: int(x.split(":")[1]) for x in buf.value.decode().split(";") if x}

    def fallback_convs(self, device=None):
        """Names (state_dict prefixes) of the convs that even per-row factors cannot fit into two fp16 pieces and which the library
        therefore serves through the three-bf16-piece kernel whatever ``gemm`` says (``dbfr_model_fallback_convs``).  None since
        ABI 5, unless a bias dwarfs its weight row by more than 2^48."""
        buf = C.create_string_buffer(4096)
        n = L.load().dbfr_model_fallback_convs(self.handle(device), buf, 4096)
        if n < 0:
            L.check(n)
        return [x for x in buf.value.decode().split(";") if x]

    def edge_log(self, device, n_steps, G):
        """Switch the per-graph edge-count read-out on for the sampler calls that follow on ``device``
        (``dbfr_model_set_edge_log``): returns the device int32 tensor [n_steps, 6, G] the library fills -- sets in the order
        (ligand, pocket, cross lig<-atom, cross atom<-lig, ligand torsion, side-chain torsion).  ``n_steps = 0`` switches it off."""
        dev = torch.device(device)
        if not isinstance(getattr(self, "_edge_log", None), dict):
            self._edge_log = {}                              # per device: every handle writes into its own tensor
        key = (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))
        if n_steps <= 0:
            L.check(L.load().dbfr_model_set_edge_log(self.handle(dev), None, 0, 0))
            self._edge_log.pop(key, None)
            return None
        log = torch.zeros(n_steps, 6, G, dtype=torch.int32, device=dev)
        L.check(L.load().dbfr_model_set_edge_log(self.handle(dev), C.c_void_p(log.data_ptr()), n_steps, G))
        self._edge_log[key] = log          # (kept alive while the library holds the pointer; a re-packed handle gets it again: handle())
        return log

    def tie_log(self, device, n_steps, G, tol=1e-5):
        """The companion of ``edge_log`` (``dbfr_model_set_tie_log``): device int32 tensor [n_steps, 6, G] = per step, edge set and graph the number of
        candidate pairs within ``tol`` Angstrom of the set's hard cutoff -- where two runs that differ by rounding (another GEMM mode, the reference
        on another batch) may build different graphs.  All-zero rows: the step's graphs are decided by margins above ``tol``.  ``n_steps = 0``
        switches it off."""
        dev = torch.device(device)
        if not isinstance(getattr(self, "_tie_log", None), dict):
            self._tie_log = {}
        key = (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))
        if n_steps <= 0:
            L.check(L.load().dbfr_model_set_tie_log(self.handle(dev), None, 0, 0, 0.0))
            self._tie_log.pop(key, None)
            return None
        log = torch.zeros(n_steps, 6, G, dtype=torch.int32, device=dev)
        L.check(L.load().dbfr_model_set_tie_log(self.handle(dev), C.c_void_p(log.data_ptr()), n_steps, G, float(tol)))
        self._tie_log[key] = (log, float(tol))
        return log

    def set_gemm(self, mode):
        """Switch the GEMM mode ("f32" | "split_f16" | "reduce_first" | None = library default at the next re-pack) of this model on every device."""
        assert mode in (None,) + tuple(GEMM_MODES)
        self.gemm = mode
        if mode is not None:
            for _, h in self._handles.values():
                L.check(L.load().dbfr_model_set_gemm(h, GEMM_MODES[mode]))

    def gemm_mode(self, device=None):
        return {v: k for k, v in GEMM_MODES.items()}[L.load().dbfr_model_get_gemm(self.handle(device))]

    def workspace(self, batch, device):
        lib = L.load()
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        nbytes = C.c_size_t()
        L.check(lib.dbfr_workspace_bytes(self.handle(dev), C.byref(batch.c), C.byref(self.limits), C.byref(nbytes)))
        ws = self._ws.get(idx)
        if ws is None or ws.numel() < nbytes.value:
            ws = self._ws[idx] = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        return ws

    def workspace_of(self, device):
        """The workspace last used on ``device`` (for ``dbfr_status_sync`` / edge counters)."""
        dev = torch.device(device)
        return self._ws[dev.index if dev.index is not None else torch.cuda.current_device()]

    @staticmethod
    def _device_of(data):
        for k in ("lig_pos", "batch", "lig_node_batch"):
            if _has(data, k):
                return _get(data, k).device
        raise KeyError("lig_pos")

    @torch.no_grad()
    def score_packed(self, pb, t, tr_sigma, rot_score_norm, tor_score_norm2, sc_tor_score_norm2, sync=True):
        dev = pb.lig_pos.device
        if dev.type != "cuda":
            raise L.DbfrError("TensorProductModelHIP needs a ROCm device (no CPU path)")
        lib = L.load()
        f = lambda x: x.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        t, tr_sigma, rot_score_norm = f(t), f(tr_sigma), f(rot_score_norm)
        tor_n2 = f(tor_score_norm2) if pb.dims["NTOR"] else torch.zeros(1, device=dev)
        sc_n2 = f(sc_tor_score_norm2) if pb.dims["NSC"] else torch.zeros(1, device=dev)
        G = pb.G
        out = dict(tr=torch.empty(G, 3, device=dev), rot=torch.empty(G, 3, device=dev),
                   tor=torch.empty(max(pb.dims["NTOR"], 1), device=dev), sc=torch.empty(max(pb.dims["NSC"], 1), device=dev))
        cond = L.Cond(*(C.c_void_p(x.data_ptr()) for x in (t, tr_sigma, rot_score_norm, tor_n2, sc_n2)))
        sc = L.Scores(*(C.c_void_p(out[k].data_ptr()) for k in ("tr", "rot", "tor", "sc")))
        with torch.cuda.device(dev):            # streams, events and allocations of the library follow the current device
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            while True:
                ws = self.workspace(pb, dev)
                L.check(lib.dbfr_score(self.handle(dev), C.byref(pb.c), C.byref(cond), C.byref(sc), C.c_void_p(ws.data_ptr()),
                                       ws.numel(), C.byref(self.limits), stream))
                if not sync:
                    break
                rc = lib.dbfr_status_sync(C.c_void_p(ws.data_ptr()), stream, None)
                if rc == L.DBFR_ERR_CAPACITY and self.auto_grow:
                    self.grow_limits(pb, ws, stream)      # the edge budgets were too small: re-plan with the counted sizes
                    continue
                L.check(rc)
                break
        return out["tr"], out["rot"], out["tor"][:pb.dims["NTOR"]], out["sc"][:pb.dims["NSC"]]

    def forward(self, data):
        """tpscore.py:462-573.  ``data``: the batched EasyDict after set_time (scFlex.py:104-122)."""
        dev = self._device_of(data)
        pb = PackedBatch(data, dev)
        scn = _get(data, "sc_tor_score_norm2")
        scm = _get(data, "sc_torsion_edge_mask").bool()
        if scn.shape == scm.shape:
            scn = scn[scm.to(scn.device)]
        tr, rot, tor, sc = self.score_packed(pb, _get(data, "t"), _get(data, "tr_sigma"), _get(data, "rot_score_norm"),
                                             _get(data, "tor_score_norm2"), scn)
        if self.no_sc_torsion:
            return tr, rot, tor, None
        return tr, rot, tor, sc
