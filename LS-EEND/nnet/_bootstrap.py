"""Puts the repository root on sys.path so the drop-in shims can import ``fs_eend_amd``."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)
