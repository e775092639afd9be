"""Drop-in import path: masked -> streaming parameter transfer (FS-EEND/streaming_infer_dia.py:16)."""
from .. import _bootstrap  # noqa: F401
from fs_eend_amd.fs_stream import (copy_params_from_masked_to_streaming, copy_params_with_conv1d,  # noqa: F401
                                   copy_params_with_masked_decoder, copy_params_with_masked_emb_encoder)
