"""Pins what CAN be pinned of the feature front-end -- build container only.

The reference's `datasets/feature.py` makes two librosa calls (`librosa.stft`, `librosa.filters.mel`); librosa and soundfile are absent
from this image, so the file is imported here with stub modules whose `filters.mel` / `stft` are INJECTED from oracle/feature_ref.py.
Everything else then runs as the reference wrote it: `transform` (the |.|^2 . mel^T, log10(max(., 1e-10)), cumulative-mean and mean
branches, feature.py:44-130), `subsample` (:133-138), `splice` (:141-163), the frame-drop rule of `stft` (:166-191) and `_count_frames`.
The fixtures therefore pin 4 of the 6 stages to outputs of the reference's own function bodies; the two injected tables stay
"parity unpinned" (see oracle/feature_ref.py), and are stored with the fixture so that a later librosa can be compared against them.

    python oracle/gen_golden_feature.py        -> tests/golden/feature_*.npz (inputs are regenerated from seeds by the test)
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import feature_ref as F        # noqa: E402

REF = "/root/reference/LS-EEND"


def import_reference_feature():
    librosa = types.ModuleType("librosa")
    librosa.filters = types.ModuleType("librosa.filters")
    librosa.filters.mel = lambda sr=8000, n_fft=256, n_mels=23, **kw: F.mel_filterbank(sr, n_fft, n_mels)      # injected table

    def _stft(data, n_fft=256, win_length=200, hop_length=80, **kw):
        # injected: the centred STFT WITHOUT the reference's own frame-drop rule (librosa returns 1 + len // hop frames, (bins, frames))
        d = np.asarray(data, dtype=np.float32)
        Y = F.stft(np.concatenate([d, np.zeros(1, np.float32)]) if len(d) % hop_length == 0 else d, win_length, hop_length)
        return Y[:1 + len(d) // hop_length].T
    librosa.stft = _stft
    sys.modules["librosa"], sys.modules["librosa.filters"] = librosa, librosa.filters
    sys.modules["soundfile"] = types.ModuleType("soundfile")
    import importlib.util                     # by path: `datasets` on sys.path is the HuggingFace package
    spec = importlib.util.spec_from_file_location("ref_ls_feature", os.path.join(REF, "datasets", "feature.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = import_reference_feature()
    out = os.path.join(ROOT, "tests", "golden")
    cases = [dict(name="feature_a", n=16000, seed=11, ctx=7, sub=10), dict(name="feature_b", n=8123, seed=12, ctx=3, sub=4),
             dict(name="feature_short", n=79, seed=13, ctx=7, sub=10), dict(name="feature_drop", n=80 * 57, seed=14, ctx=7, sub=10)]
    for c in cases:
        g = np.random.default_rng(c["seed"])
        y = (g.standard_normal(c["n"]) * 0.05).astype(np.float32)
        Y = ref.stft(y, 200, 80)                                     # reference's wrapper (frame-drop rule) around the injected STFT
        arrays = dict(stft_frames=np.array([Y.shape[0], ref._count_frames(len(y), 200, 80)], dtype=np.int64))
        feats = {}
        for tt in ("logmel23", "logmel23_mn", "logmel23_cummn"):
            feats[tt] = ref.transform(Y, tt)
            arrays["transform_" + tt] = feats[tt]
        lab = (g.random((Y.shape[0], 3)) > 0.5).astype(np.int32)
        sp = np.ascontiguousarray(ref.splice(feats["logmel23_cummn"], c["ctx"]))
        ys, ts = ref.subsample(sp, lab, c["sub"])
        arrays.update(splice=sp[:: max(1, sp.shape[0] // 16)], splice_shape=np.array(sp.shape), subsample_Y=np.ascontiguousarray(ys),
                      subsample_T=np.ascontiguousarray(ts), mel_injected=F.mel_filterbank(), input_dim=np.array([ref.get_input_dim(200, c["ctx"], "logmel23_cummn")]))
        meta = np.array([c["n"], c["seed"], c["ctx"], c["sub"]], dtype=np.int64)
        np.savez_compressed(os.path.join(out, c["name"] + ".npz"), meta=meta, **arrays)
        print(c["name"], {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main()
