"""Load the reference's own Python sources in the BUILD CONTAINER ONLY.

Used by tests/golden/make_golden.py (and tests/golden/make_residue_tables.py) to pin
the oracle against the reference implementation and to freeze golden vectors.
/root/reference does not exist on the GPU box and nothing under tests/ that
runs there imports this module.

Recipe: a writable copy of the reference, bare parent
packages (``__init__`` not executed) so leaf modules import by path, and tiny
stand-ins for the third-party modules that are absent offline.  The e3nn /
torch_cluster / torch_scatter stand-ins are the oracle's own restatements
(oracle/e3nn_lite.py, oracle/cluster.py): at that boundary parity stays
UNPINNED -- what this harness pins is everything the reference itself wrote
(tpscore.py / scFlex.py glue, geometry, schedules, embeddings, LayerNorm).
"""
import importlib
import importlib.util
import os
import shutil
import sys
import types

REF = "/root/reference"
COPY = "/tmp/ref_copy"


def available():
    return os.path.isdir(os.path.join(REF, "druglib"))


def _bare(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


class _Anything(types.ModuleType):
    """Attribute-returning stub for unused third-party imports."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        sub = _Anything(f"{self.__name__}.{k}")
        sys.modules[sub.__name__] = sub
        setattr(self, k, sub)
        return sub

    def __call__(self, *a, **k):
        return self


def install():
    """Idempotent; returns the root of the writable copy."""
    if "druglib" in sys.modules and getattr(sys.modules["druglib"], "_shimmed", False):
        return COPY
    assert available(), "reference not mounted (this only runs in the build container)"
    if not os.path.isdir(COPY):
        shutil.copytree(REF, COPY, ignore=shutil.ignore_patterns("*.pdb", "*.sdf", "*.ipynb", "images", "notebooks"))
        os.system(f"chmod -R u+w {COPY}")
    sys.path.insert(0, COPY)
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    import torch
    from oracle import cluster, e3nn_lite

    d = os.path.join(COPY, "druglib")
    root = _bare("druglib", d)
    root._shimmed = True
    for sub in ["utils", "utils/obj", "utils/bio_utils", "utils/geometry_utils", "data", "core", "models",
                "models/Docking", "models/Docking/interaction", "models/Docking/encoder", "models/Base",
                "models/Base/diffusion", "apis", "datasets", "datasets/Docking"]:
        _bare("druglib." + sub.replace("/", "."), os.path.join(d, sub))

    # ---- third-party stand-ins
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setattr__(k, v)
            super().__setitem__(k, v)

        __setitem__ = __setattr__

        def pop(self, k, *a):
            if hasattr(self, k):
                delattr(self, k)
            return super().pop(k, *a)

    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    def map_structure(fn, s):
        if isinstance(s, (list, tuple)):
            return type(s)(map_structure(fn, x) for x in s)
        if isinstance(s, dict):
            return {k: map_structure(fn, v) for k, v in s.items()}
        return fn(s)

    tr = types.ModuleType("tree")
    tr.map_structure = map_structure
    sys.modules["tree"] = tr
    for nm in ["lmdb", "rdkit", "rdkit.Chem", "rdkit.Chem.AllChem", "rdkit.Geometry", "torch_sparse",
               "torch_geometric", "torch_geometric.nn", "torch_geometric.typing", "torch_geometric.utils",
               "torch_geometric.data", "networkx", "tqdm", "prody", "Bio", "Bio.PDB", "openmm"]:
        if nm == "networkx" or nm == "tqdm":
            try:
                importlib.import_module(nm)
                continue
            except Exception:
                pass
        if nm not in sys.modules:
            sys.modules[nm] = _Anything(nm)
    ts = types.ModuleType("torch_scatter")
    ts.scatter, ts.scatter_add, ts.scatter_mean = cluster.scatter, cluster.scatter_add, cluster.scatter_mean
    ts.scatter_sum = cluster.scatter_add
    sys.modules["torch_scatter"] = ts
    tc = types.ModuleType("torch_cluster")
    tc.radius, tc.radius_graph = cluster.radius, cluster.radius_graph
    sys.modules["torch_cluster"] = tc
    e3 = types.ModuleType("e3nn")
    e3.o3 = e3nn_lite
    sys.modules["e3nn"] = e3
    sys.modules["e3nn.o3"] = e3nn_lite

    # ---- druglib internals the hot-path files import from package __init__s
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(d, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    import torch.nn as nn

    apis = sys.modules["druglib.apis"]
    apis.get_activation = lambda name: {"relu": nn.ReLU, "tanh": nn.Tanh}[name.lower()]

    def _xavier(m, gain=1, bias=0, distribution="normal"):
        if hasattr(m, "weight") and m.weight is not None:
            (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(m.weight, gain=gain)
        if hasattr(m, "bias") and m.bias is not None:
            nn.init.constant_(m.bias, bias)

    apis.xavier_init = apis.kaiming_init = apis.glorot_init = _xavier

    # registry: a faithful-enough mmcv-style stand-in (the real one drags in the
    # whole druglib.utils tree: config/addict/yapf/...)
    class Registry:
        def __init__(self, name, **kw):
            self.name, self._d = name, {}

        def register_module(self, name=None, force=False, module=None, **kw):
            def deco(cls):
                self._d[name or cls.__name__] = cls
                return cls
            return deco(module) if module is not None else deco

        def get(self, k):
            return self._d.get(k)

        def build(self, cfg, **kw):
            cfg = dict(cfg)
            t = cfg.pop("type")
            cfg.update(kw.get("default_args") or {})
            return self._d[t](**cfg)

    utils = sys.modules["druglib.utils"]
    utils.Registry = Registry
    utils.build_from_cfg = lambda cfg, reg, default_args=None: reg.build(cfg, default_args=default_args)
    tu = importlib.import_module("druglib.utils.torch_utils")
    go = sys.modules["druglib.utils.geometry_utils"]
    gu = load("druglib.utils.geometry_utils.utils", "utils/geometry_utils/utils.py")
    for k in dir(gu):
        if not k.startswith("_"):
            setattr(go, k, getattr(gu, k))
    sup = load("druglib.utils.geometry_utils.superimposition", "utils/geometry_utils/superimposition.py")
    go.rigid_transform_Kabsch_3D_torch = sup.rigid_transform_Kabsch_3D_torch
    obj = sys.modules["druglib.utils.obj"]
    pc = load("druglib.utils.obj.protein_constants", "utils/obj/protein_constants.py")
    obj.protein_constants = pc
    af = load("druglib.utils.geometry_utils.aaframe", "utils/geometry_utils/aaframe.py")
    go.aaframe = af
    return COPY


def mod(name):
    install()
    return importlib.import_module(name)


def _load(name, rel):
    d = os.path.join(COPY, "druglib")
    if name in sys.modules and getattr(sys.modules[name], "__file__", None):
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_hot_path():
    """Import every reference module on the hot path; returns a namespace of them."""
    install()
    import torch.nn as nn
    ns = types.SimpleNamespace()
    ns.pc = sys.modules["druglib.utils.obj.protein_constants"]
    ns.geom = sys.modules["druglib.utils.geometry_utils.utils"]
    ns.superimposition = sys.modules["druglib.utils.geometry_utils.superimposition"]
    ns.aaframe = sys.modules["druglib.utils.geometry_utils.aaframe"]
    ns.torch_utils = importlib.import_module("druglib.utils.torch_utils")
    ns.conformer = _load("druglib.utils.bio_utils.conformer_utils", "utils/bio_utils/conformer_utils.py")
    bio = sys.modules["druglib.utils.bio_utils"]
    bio.update_batchlig_pos = ns.conformer.update_batchlig_pos
    bio.modify_conformer_torsion_angles = ns.conformer.modify_conformer_torsion_angles
    ns.prot_math = _load("druglib.utils.obj.prot_math", "utils/obj/prot_math.py")
    sys.modules["druglib.utils.obj"].build_pdb_from_template = ns.prot_math.build_pdb_from_template
    ns.time_emb = _load("druglib.models.Base.diffusion.time_emb", "models/Base/diffusion/time_emb.py")
    ns.schnet = _load("druglib.models.Docking.interaction.schnet", "models/Docking/interaction/schnet.py")
    ns.encoder = _load("druglib.models.Docking.encoder.equibind_encoder", "models/Docking/encoder/equibind_encoder.py")
    utils = sys.modules["druglib.utils"]
    utils.Config = dict
    core = sys.modules["druglib.core"]

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    core.BaseModule = BaseModule
    core.auto_fp16 = lambda *a, **k: (lambda f: f)
    sys.modules["druglib.data"].BaseData = object
    _load("druglib.models.base_model_builder", "models/base_model_builder.py")
    ns.builder = _load("druglib.models.builder", "models/builder.py")
    ns.tpscore = _load("druglib.models.Docking.interaction.tpscore", "models/Docking/interaction/tpscore.py")
    _load("druglib.models.Docking.default_MLDockBuilder", "models/Docking/default_MLDockBuilder.py")
    _load("druglib.models.Docking.base", "models/Docking/base.py")
    ns.scflex = _load("druglib.models.Docking.scFlex", "models/Docking/scFlex.py")
    ns.EasyDict = sys.modules["easydict"].EasyDict
    return ns
