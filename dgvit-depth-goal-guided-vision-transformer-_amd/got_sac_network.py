"""Drop-in name for the reference module: ``from dgvit_amd.got_sac_network import GoTPolicy, GoTQNetwork, ...``."""
from .sac_networks import (GoTPolicy, GoTQNetwork, DeterministicGoTPolicy, weights_init_, set_seed,  # noqa: F401
                           LOG_SIG_MAX, LOG_SIG_MIN, epsilon)
from .cnn_networks import QNetwork, GaussianPolicy  # noqa: F401,E402
