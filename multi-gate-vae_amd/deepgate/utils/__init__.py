from .utils import zero_normalization, AverageMeter  # noqa: F401
from .logger import Logger  # noqa: F401
from .model_utils import load_model  # noqa: F401
