"""Drop-in import path for the layer-stack container and the time x speaker fusion layer."""
from .. import _bootstrap  # noqa: F401
from fs_eend_amd.fs_model import TransformerEncoder, TransformerEncoderFusionLayer  # noqa: F401
