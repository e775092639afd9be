"""Drop-in import path: `from train.utils.make_rttm import make_rttm` (dia_pred.py:13, streaming_infer_dia.py).
threshold -> median filter -> change points on the device (csrc/postproc.hip); accepts the CPU `pred` tensors the
reference's drivers pass."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)
from fs_eend_amd.postproc import make_rttm  # noqa: E402,F401
