"""Import alias: the package directory is ``fs-eend_amd/`` (not a valid Python
identifier), so ``import fs_eend_amd`` resolves here and this module turns itself
into a package whose submodules live in that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "fs-eend_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
