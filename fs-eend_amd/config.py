"""YAML configs with `!ref <section[key]>` references, without hyperpyyaml (not installed here; SURVEY 2 last row).

The reference's configs (FS-EEND/conf/*.yaml, LS-EEND/conf/*.yaml) use two hyperpyyaml features only:
`!ref <a[b]>` as a whole value (the referenced value, typed) and `!ref text<a[b]>text` (string interpolation).
`load()` resolves exactly those; everything else is plain YAML.  `FS_EEND_SIMU` / `LS_EEND_SIMU` are this repository's
own tables of the values the hot path needs (shapes, optimiser), read by bench.py, the golden generators and the tests.
"""
import os
import re

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
FS_EEND_SIMU = os.path.join(_HERE, "conf", "fs_eend_simu.yaml")
LS_EEND_SIMU = os.path.join(_HERE, "conf", "ls_eend_simu.yaml")
_REF = re.compile(r"<([^<>]+)>")


class _Ref(str):
    pass


class _Loader(yaml.SafeLoader):
    pass


_Loader.add_constructor("!ref", lambda loader, node: _Ref(loader.construct_scalar(node)))


def _lookup(root, path: str):
    """`a[b][c]` -> root['a']['b']['c'] (integer keys for lists)."""
    head, *rest = re.findall(r"[^\[\]]+", path)
    cur = root[head]
    for k in rest:
        cur = cur[int(k)] if isinstance(cur, list) else cur[k]
    return cur


def _resolve(node, root, depth=0):
    if depth > 32:
        raise ValueError("!ref cycle")
    if isinstance(node, _Ref):
        m = _REF.fullmatch(node.strip())
        if m:                                              # whole-value reference keeps the referenced type
            return _resolve(_lookup(root, m.group(1)), root, depth + 1)
        return _REF.sub(lambda mm: str(_resolve(_lookup(root, mm.group(1)), root, depth + 1)), str(node))
    if isinstance(node, dict):
        return {k: _resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    return node


def loads(text: str) -> dict:
    raw = yaml.load(text, Loader=_Loader)
    return _resolve(raw, raw)


def load(path: str) -> dict:
    with open(path) as f:
        return loads(f.read())


def model_kwargs(cfg: dict) -> dict:
    """`Model(n_speakers=..., in_size=..., **configs["model"]["params"])` (FS-EEND/train_dia.py:69-73)."""
    d = cfg["data"]
    return dict(n_speakers=d.get("num_speakers"), in_size=(2 * d["context_recp"] + 1) * d["feat"]["n_mels"], **cfg["model"]["params"])
