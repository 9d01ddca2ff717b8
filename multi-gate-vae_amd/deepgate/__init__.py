"""MI355X-native drop-in for the `deepgate` package of 959AI994/Multi-Gate-VAE (DG_AE hot path).

Same names as the reference's package (DG_VAE/deepgate/__init__.py:1-10): `deepgate.Model` is the
last `Model` imported there, i.e. the XAG one; `train.py` picks the per-type class explicitly."""
from . import synthetic  # noqa: F401
from . import digae_layer, digvae_model  # noqa: F401
from . import dg_ae_model_aig, dg_ae_model_mig, dg_ae_model_xmg, dg_ae_model_xag  # noqa: F401
from .dg_ae_model_xag import Model  # noqa: F401
from .trainer import Trainer, GraphLoader  # noqa: F401
from .data import CircuitBatch  # noqa: F401
from .parser import NpzParser  # noqa: F401
from .graph_plan import GraphPlan  # noqa: F401
from .optim import FlatAdam  # noqa: F401
from .utils import zero_normalization, AverageMeter  # noqa: F401
from .__version__ import __version__  # noqa: F401
