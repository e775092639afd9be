"""Making the reference's entry scripts bind to this implementation.

The reference runs its scripts with the project root as CWD, so `nnet`, `train`, `datasets` resolve as top-level
*namespace* packages (no __init__.py anywhere in the reference tree).  The shim trees `FS-EEND/` and `LS-EEND/` of this
repository are namespace portions too: when the shim directory comes BEFORE the reference's project root on
`sys.path`, Python merges both portions, every module the shim provides (`nnet.model.*`, `train.utils.loss`, ...)
wins, and everything it does not provide (`train.oln_tfm_enc_dec`, `datasets.diarization_dataset`, `utlis.*`, ...)
still resolves to the reference's own file.  `python script.py` puts the script's directory at sys.path[0], in front of
PYTHONPATH, so a plain PYTHONPATH override cannot work; `run()` (python -m fs_eend_amd.run) sets the order up correctly
and then executes the script as __main__.

`overlay()` lets a shim module that replaces only SOME names of a reference module (e.g. train/utils/loss.py) start
from the reference's own definitions: it executes the same-named file of a later namespace portion into the shim's
namespace first, then the shim overrides what it accelerates.
"""
import os
import runpy
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def shim_dir(flavour: str) -> str:
    if flavour not in ("FS-EEND", "LS-EEND"):
        raise ValueError("flavour must be 'FS-EEND' or 'LS-EEND'")
    return os.path.join(ROOT, flavour)


def overlay(ns: dict, modname: str, shim_file: str):
    """Execute the reference's module `modname` (a later portion of the merged namespace package) into `ns`.
    Returns the path executed, or None when there is no other portion (stand-alone use of the shim tree) or the
    reference module's own dependencies are missing (recorded in ns['__overlay_error__'])."""
    pkg_name, _, leaf = modname.rpartition(".")
    pkg = sys.modules.get(pkg_name)
    for d in list(getattr(pkg, "__path__", [])):
        cand = os.path.join(d, leaf + ".py")
        if os.path.exists(cand) and not os.path.samefile(cand, shim_file):
            try:
                with open(cand) as f:
                    exec(compile(f.read(), cand, "exec"), ns)
            except ImportError as e:                     # e.g. torchmetrics / h5py / librosa absent on this box
                ns["__overlay_error__"] = e
                return None
            return cand
    return None


def install_namespaces(dirs):
    """Pre-register the merged top-level namespace packages (`nnet`, `train`, `datasets`, ...) over `dirs` (in
    priority order).  Needed because a REGULAR package of the same name anywhere on sys.path beats namespace portions
    (e.g. the Hugging Face `datasets` distribution in site-packages shadows the reference's `datasets/` directory,
    which has no __init__.py); an explicit sys.modules entry is immune to that.  Sub-packages (`train.utils`,
    `nnet.model`) then merge by themselves through the parent's __path__."""
    import importlib.machinery
    import types
    names = []
    for d in dirs[:1]:
        names = sorted(n for n in os.listdir(d) if os.path.isdir(os.path.join(d, n)) and not n.startswith(("_", ".")))
    for name in names:
        portions = [os.path.join(d, name) for d in dirs if os.path.isdir(os.path.join(d, name))]
        mod = types.ModuleType(name)
        mod.__path__ = portions
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        mod.__spec__.submodule_search_locations = portions
        sys.modules[name] = mod
    return names


def arrange_sys_path(script: str, flavour: str = None):
    """sys.path = [shim dir, script dir, repo root, ...rest] and the merged namespaces installed;
    returns (shim dir, script dir)."""
    script_dir = os.path.dirname(os.path.abspath(script))
    if flavour is None:
        parts = script_dir.split(os.sep)
        flavour = next((p for p in reversed(parts) if p in ("FS-EEND", "LS-EEND")), None)
        if flavour is None:
            raise SystemExit("cannot tell FS-EEND from LS-EEND from the script path; pass --flavour")
    sd = shim_dir(flavour)
    rest = [p for p in sys.path if p and os.path.abspath(p) not in (sd, script_dir, ROOT)]
    sys.path[:] = [sd, script_dir, ROOT] + rest
    install_namespaces([sd, script_dir])
    return sd, script_dir


def run(argv=None):
    """python -m fs_eend_amd.run [--flavour FS-EEND|LS-EEND] <reference script.py> [script args...]"""
    argv = list(sys.argv[1:] if argv is None else argv)
    flavour = None
    if argv[:1] == ["--flavour"]:
        flavour, argv = argv[1], argv[2:]
    if not argv:
        raise SystemExit(run.__doc__)
    script = argv[0]
    arrange_sys_path(script, flavour)
    os.chdir(os.path.dirname(os.path.abspath(script)))      # the reference resolves conf/, example/ relative to its root
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
