"""`Model` for XAG circuits — drop-in for DG_VAE/deepgate/dg_ae_model_xag.py.
Gate ids: XAG: NOT 2 / AND 3 / XOR 5 (dg_ae_model_xag.py:81-83)."""
from ._model_base import FunctionalModel, EPS, MAX_LOGSTD  # noqa: F401


class Model(FunctionalModel):
    ENCODER_ATTR = 'xag_struct_encoder'
    GATES = (('and', 3), ('not', 2), ('xor', 5))
