"""fs-eend_amd -- MI355X (gfx950) native FS-EEND / LS-EEND frame-wise diarization
hot path: hand-written HIP kernels behind a C-ABI (include/eend_hip.h), a ctypes
binding (lib.py, ops.py) and host-side mirrors of the reference's nnet/ modules
(fs_model.py, fs_streaming.py, ...).  Import as ``fs_eend_amd``."""
__version__ = "0.1.0"
