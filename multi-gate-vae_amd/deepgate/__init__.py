"""MI355X-native drop-in for the `deepgate` package of 959AI994/Multi-Gate-VAE (DG_AE hot path)."""
from . import synthetic  # noqa: F401
from . import digae_layer  # noqa: F401
from .graph_plan import GraphPlan  # noqa: F401
from .__version__ import __version__  # noqa: F401
