"""Drop-in for the reference module of the same path (LS-EEND/train_dia_simu.py and
LS-EEND/streaming_infer_dia.py:14-17 import from here).  Implementation: MI355X HIP path."""
from .. import _bootstrap  # noqa: F401
from fs_eend_amd.ls_model import (EmbeddingEncoderModule, MaskedTransformerDecoderModel,  # noqa: F401
                                  OnlineConformerRetentionDADiarization, StreamingConv1d)
from fs_eend_amd.fs_model import PositionalEncoding  # noqa: F401
