"""Drop-in import path for the frame-level DER report (`train/utils/loss.py`: calc_diarization_error,
report_diarization_error).  The training losses of that module (PIT) are not part of this build."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from fs_eend_amd.postproc import calc_diarization_error, report_diarization_error  # noqa: E402,F401
