"""Drop-in for the reference's `train/utils/loss.py`.  Starts from the reference's own definitions when its file is
present on the merged namespace path (so `standard_loss`, `batch_pit_loss`, ... keep existing for
`train/oln_tfm_enc_dec*.py`), then replaces what this build accelerates: the frame-level DER report and the
permutation-invariant label assignment (device kernels, no host round trip)."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)
from fs_eend_amd.dropin import overlay  # noqa: E402

_REFERENCE_FILE = overlay(globals(), __name__, __file__)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

if "standard_loss" not in globals():
    def standard_loss(ys, ts, label_delay=0):
        """train/utils/loss.py:119-125 (stand-alone use of the shim tree, reference file absent)."""
        losses = [F.binary_cross_entropy_with_logits(y[label_delay:, ...], t[:len(t) - label_delay, ...]) * (len(y) - label_delay)
                  for (y, t) in zip(ys, ts)]
        n_frames = sum(t.shape[0] for t in ts) - label_delay * len(ts)
        return torch.stack(losses).sum() / n_frames

from fs_eend_amd.postproc import calc_diarization_error, report_diarization_error  # noqa: E402,F401
from fs_eend_amd.pit import batch_pit_n_speaker_loss, pad_labels, pad_preds, pit_loss_multispk  # noqa: E402,F401
