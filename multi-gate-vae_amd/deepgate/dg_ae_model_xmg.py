"""`Model` for XMG circuits — drop-in for DG_VAE/deepgate/dg_ae_model_xmg.py.
Gate ids: XMG: MAJ 1 / NOT 2 / AND 3 / OR 4 / XOR 5 (dg_ae_model_xmg.py:86-90)."""
from ._model_base import FunctionalModel, EPS, MAX_LOGSTD  # noqa: F401


class Model(FunctionalModel):
    ENCODER_ATTR = 'xmg_struct_encoder'
    GATES = (('and', 3), ('not', 2), ('xor', 5), ('maj', 1), ('or', 4))
