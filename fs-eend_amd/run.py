"""python -m fs_eend_amd.run <reference script> ...  -- see dropin.run()."""
from .dropin import run

if __name__ == "__main__":
    run()
