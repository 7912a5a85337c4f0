"""Drop-in name for the reference module: ``from dgvit_amd.GoalFormer import GoT``."""
from .goalformer import GoT, Transformer, Attention, FeedForward, PreNorm, RMSNorm, pair  # noqa: F401
