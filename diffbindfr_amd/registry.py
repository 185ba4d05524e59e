"""mmcv-style registries: the reference's operator/plugin boundary for this path.

Mirrors druglib/utils/registry.py:60-358 (``Registry.register_module(name=, overwrite=, module=)``,
``Registry.build(cfg, default_args=)`` popping ``type``) and the registry objects of
druglib/models/builder.py:7-17.  When ``druglib`` itself is importable the classes of
this package are ALSO registered into the reference's own ``INTERACTION`` /
``MLDOCK_BUILDER`` so that ``--cfg-options model.type=DiffBindFRHIP`` (or
``model.diffusion_model.type=TensorProductModelHIP``) selects them from
DiffBindFR/app/predict.py with no change to the reference.
"""
import copy


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __contains__(self, key):
        return key in self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, overwrite=False):
        names = [name or cls.__name__] if not isinstance(name, (list, tuple)) else list(name)
        for n in names:
            if not overwrite and n in self._module_dict:
                raise KeyError(f"{n} is already registered, in {self._name}")
            self._module_dict[n] = cls

    def register_module(self, name=None, overwrite=False, module=None):
        """registry.py:285-340: decorator or ``register_module(module=cls)``; ``overwrite`` replaces an existing entry."""
        if not isinstance(overwrite, bool):
            raise KeyError(f'"overwrite" must be a boolean, but got a type "{type(overwrite)}"')
        if module is not None:
            self._register(module, name, overwrite)
            return module

        def deco(cls):
            self._register(cls, name, overwrite)
            return cls
        return deco

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise KeyError("`cfg` must be a dict containing the key 'type'")
        args = copy.copy(dict(cfg))
        typ = args.pop("type")
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self._name} registry")
        return cls(**args)


INTERACTION = Registry("interaction")
ENERGY = Registry("energy")
MLDOCK_BUILDER = Registry("mldock model builder")


def build_interaction(cfg):
    """druglib/models/builder.py:32-34."""
    return INTERACTION.build(cfg)


def build_energy(cfg):
    """druglib/models/builder.py:36-38 (the registry the reference's ``scoring_model=`` goes through; it registers nothing
    there itself -- its MDN scorer lives in DiffBindFR/scoring and is driven by common/engines.py:Scorer)."""
    return ENERGY.build(cfg)


def register_into_druglib():
    """Best effort: plug into the reference's registries when it is importable."""
    try:
        from druglib.models.builder import INTERACTION as REF_I, MLDOCK_BUILDER as REF_M, ENERGY as REF_E  # noqa
    except Exception:
        return False
    for reg, ref in ((INTERACTION, REF_I), (MLDOCK_BUILDER, REF_M), (ENERGY, REF_E)):
        for n, cls in reg.module_dict.items():
            if ref.get(n) is not cls:
                ref.register_module(name=n, overwrite=ref.get(n) is not None, module=cls)
    # ``isinstance(model, BaseMLDocker)`` gates in a user's fork: the drop-in is registered as a VIRTUAL subclass (the reference's base is an
    # ABCMeta class, druglib/models/Docking/base.py:13) -- it has the inference surface, not the training runner's
    try:
        from druglib.models.Docking.base import BaseMLDocker
        for n, cls in MLDOCK_BUILDER.module_dict.items():
            BaseMLDocker.register(cls)
    except Exception:
        pass
    return True
