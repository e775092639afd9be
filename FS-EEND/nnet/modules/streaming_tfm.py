"""Drop-in import path: FS-EEND frame-by-frame streaming modules."""
from .. import _bootstrap  # noqa: F401
from fs_eend_amd.fs_stream import (IncrementalSelfAttention, StreamingAttractorDecoder,  # noqa: F401
                                   StreamingAttractorDecoderLayer, StreamingConv1d, StreamingEmbeddingEncoder,
                                   StreamingTransformerEncoderLayer)
