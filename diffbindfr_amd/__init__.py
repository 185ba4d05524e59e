"""diffbindfr_amd -- MI355X-native reverse-diffusion sampler for DiffBindFR's pose-denoising
hot path (hand-written HIP for gfx950 behind the reference's registry boundary).

Importing the package registers ``TensorProductModelHIP`` (INTERACTION) and
``DiffBindFRHIP`` (MLDOCK_BUILDER); when the reference's ``druglib`` is importable they are
registered into its registries as well.
"""
from .registry import INTERACTION, MLDOCK_BUILDER, register_into_druglib  # noqa: F401
from .score_model import TensorProductModelHIP  # noqa: F401
from .sampler import DiffBindFRHIP  # noqa: F401

register_into_druglib()
