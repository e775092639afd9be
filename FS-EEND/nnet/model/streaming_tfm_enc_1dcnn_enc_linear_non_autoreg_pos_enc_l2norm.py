"""Drop-in for the reference module of the same path (FS-EEND/streaming_infer_dia.py:13-15)."""
from .. import _bootstrap  # noqa: F401
from fs_eend_amd.fs_stream import StreamingTransformerEDADiarization  # noqa: F401
