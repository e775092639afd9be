"""Drop-in for the reference module of the same path (FS-EEND/train_dia.py:22 imports
``OnlineTransformerDADiarization`` from here).  The implementation is the MI355X HIP path."""
from .. import _bootstrap  # noqa: F401
from fs_eend_amd.fs_model import (MaskedTransformerDecoderModel, MaskedTransformerEncoderModel,  # noqa: F401
                                  OnlineTransformerDADiarization, PositionalEncoding)
