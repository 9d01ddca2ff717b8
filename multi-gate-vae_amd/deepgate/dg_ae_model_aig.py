"""`Model` for AIG circuits — drop-in for DG_VAE/deepgate/dg_ae_model_aig.py.
Gate ids: AIG: PI 0 / AND 1 / NOT 2 (dg_ae_model_aig.py:67-68)."""
from ._model_base import FunctionalModel, EPS, MAX_LOGSTD  # noqa: F401


class Model(FunctionalModel):
    ENCODER_ATTR = 'struct_encoder'
    GATES = (('and', 1), ('not', 2))
