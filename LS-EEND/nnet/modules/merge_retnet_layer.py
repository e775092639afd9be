"""Drop-in import path: LS-EEND decoder layer (retention over time, MHA over speakers, FFN)."""
from .. import _bootstrap  # noqa: F401
from fs_eend_amd.ls_model import TransformerEncoderFusionLayer  # noqa: F401
