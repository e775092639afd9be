"""Drop-in import path: `from train.utils.make_rttm import make_rttm` (dia_pred.py:13, streaming_infer_dia.py)."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from fs_eend_amd.postproc import make_rttm  # noqa: E402,F401
