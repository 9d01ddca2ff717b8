"""`Model` for MIG circuits — drop-in for DG_VAE/deepgate/dg_ae_model_mig.py.
Gate ids: MIG: MAJ 1 / NOT 2 / AND 3 / OR 4 (dg_ae_model_mig.py:79-82)."""
from ._model_base import FunctionalModel, EPS, MAX_LOGSTD  # noqa: F401


class Model(FunctionalModel):
    ENCODER_ATTR = 'mig_struct_encoder'
    GATES = (('and', 3), ('not', 2), ('or', 4), ('maj', 1))
