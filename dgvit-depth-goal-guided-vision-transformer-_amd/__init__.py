"""dgvit_amd -- MI355X-native DGViT encoder hot path.

Python host code over the C ABI of libdgvit_hip.so (include/dgvit_hip.h).  Importing the package never
touches the GPU; the first call loads the library and raises ``DgvitError`` if it is missing.
``with dgvit_amd.diagnostic_library() as lib:`` routes the package through libdgvit_hip_diag.so (the -DDGVIT_DIAG build with the
A/B knobs and experiments of include/dgvit_hip_diag.h) for tools/ and equality tests; the product library has no knobs.
"""
from ._lib import DgvitError, LIB_PATH, DIAG_LIB_PATH, load as load_library, diagnostic as diagnostic_library  # noqa: F401
from .goalformer import GoT  # noqa: F401
from .sac_networks import GoTPolicy, GoTQNetwork, DeterministicGoTPolicy, weights_init_  # noqa: F401
from .cnn_networks import QNetwork, GaussianPolicy  # noqa: F401
from . import functional  # noqa: F401
from . import preprocess  # noqa: F401
from .runtime import GraphedStep  # noqa: F401

__all__ = ["GoT", "GoTPolicy", "GoTQNetwork", "DeterministicGoTPolicy", "QNetwork", "GaussianPolicy", "weights_init_", "functional", "GraphedStep", "DgvitError",
           "load_library", "diagnostic_library", "LIB_PATH", "DIAG_LIB_PATH"]
