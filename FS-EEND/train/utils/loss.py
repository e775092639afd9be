"""Drop-in import path for `train/utils/loss.py`: the frame-level DER report and the permutation-invariant
label assignment (values only -- the training losses are not differentiable in this build)."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from fs_eend_amd.postproc import calc_diarization_error, report_diarization_error  # noqa: E402,F401
from fs_eend_amd.pit import batch_pit_n_speaker_loss, pad_labels, pad_preds, pit_loss_multispk  # noqa: E402,F401
