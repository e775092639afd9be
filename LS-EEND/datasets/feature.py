"""Drop-in for the reference's `datasets/feature.py`: starts from the reference's own definitions when its file is on the
merged namespace path (kaldi_data / diarization_dataset keep finding `get_labeledSTFT`, `transform`, ... there; needs
librosa + soundfile like the reference), then replaces the wave -> log-mel -> splice -> subsample chain with the GPU
front-end (csrc/feature.hip)."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)
from fs_eend_amd.dropin import overlay  # noqa: E402

_REFERENCE_FILE = overlay(globals(), __name__, __file__)
from fs_eend_amd.feature import extract_fbank, extract_fbank_wave, logmel, splice, splice_subsample, subsample  # noqa: E402,F401
