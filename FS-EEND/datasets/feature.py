"""Drop-in import path: `from datasets.feature import *` (dia_pred.py:12) -- the GPU feature front-end."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from fs_eend_amd.feature import extract_fbank, extract_fbank_wave, logmel, splice, splice_subsample, subsample  # noqa: E402,F401
