"""Drop-in import path: Conformer-with-retention encoder."""
from .. import _bootstrap  # noqa: F401
from fs_eend_amd.ls_model import ConformerEncoder, ConformerEncoderBlock  # noqa: F401
