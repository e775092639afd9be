"""Drop-in import path: parameter containers of the retention module."""
from .. import _bootstrap  # noqa: F401
from fs_eend_amd.ls_model import MultiScaleRetention, RetNetRelPos  # noqa: F401
