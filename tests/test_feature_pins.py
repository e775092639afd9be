"""Feature front-end against fixtures produced by the REFERENCE's own `datasets/feature.py` (oracle/gen_golden_feature.py: the file is
imported with stub librosa / soundfile modules; its two librosa calls -- the STFT and the mel table -- are injected from
oracle/feature_ref.py and stay "parity unpinned").  Pinned by these fixtures: `transform` (power -> mel -> log10 floor -> cumulative-mean /
mean branches, LS-EEND/datasets/feature.py:44-130), `splice` (:141-163), `subsample` (:133-138), the frame-drop rule of `stft`
(:166-191), `_count_frames`, `get_input_dim`.  CPU: the oracle restatement; GPU: the HIP front-end (feature.hip)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import feature_ref as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "feature_*.npz")))


def _load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    n, seed, ctx, sub = (int(v) for v in z["meta"])
    g = np.random.default_rng(seed)
    y = (g.standard_normal(n) * 0.05).astype(np.float32)
    return z, y, ctx, sub, g


def test_fixtures_exist():
    assert len(CASES) >= 4


@pytest.mark.parametrize("name", CASES)
def test_oracle_vs_reference_functions(name):
    z, y, ctx, sub, g = _load(name)
    Y = F.stft(y, 200, 80)
    assert Y.shape[0] == int(z["stft_frames"][0]) == int(z["stft_frames"][1])          # frame-drop rule == _count_frames
    assert np.array_equal(F.mel_filterbank(), z["mel_injected"])                        # the injected table is the oracle's
    feats = {}
    for tt in ("logmel23", "logmel23_mn", "logmel23_cummn"):
        feats[tt] = F.transform(Y, tt)
        want = z["transform_" + tt]
        assert feats[tt].shape == want.shape and feats[tt].dtype == want.dtype
        assert np.abs(feats[tt] - want).max() <= 1e-6, tt                               # same numpy lines: float32 round-off only
    sp = F.splice(feats["logmel23_cummn"], ctx)
    assert tuple(z["splice_shape"]) == sp.shape and sp.shape[1] == int(z["input_dim"][0])
    step = max(1, sp.shape[0] // 16)
    assert np.abs(sp[::step] - z["splice"]).max() <= 1e-6
    lab = (g.random((Y.shape[0], 3)) > 0.5).astype(np.int32)
    assert np.abs(sp[::sub] - z["subsample_Y"]).max() <= 1e-6 and np.array_equal(lab[::sub], z["subsample_T"])
    assert np.abs(F.extract_fbank_wave(y, ctx, 200, 80, "logmel23_cummn", sub) - z["subsample_Y"]).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_front_end_vs_reference_functions(hip_lib, dev, name):
    from fs_eend_amd import feature
    z, y, ctx, sub, g = _load(name)
    yd = torch.from_numpy(y).to(dev)
    for tt in ("logmel23", "logmel23_mn", "logmel23_cummn"):
        got = feature.logmel(yd, input_transform=tt).cpu().numpy()
        want = z["transform_" + tt]
        assert got.shape == want.shape and np.abs(got - want).max() < 2e-4, tt           # fp32 DFT-by-matrix-product vs float64 FFT
    got = feature.extract_fbank_wave(yd, context_size=ctx, input_transform="logmel23_cummn", subsampling=sub).cpu().numpy()
    assert got.shape == z["subsample_Y"].shape and np.abs(got - z["subsample_Y"]).max() < 2e-4
    # splice / subsample are exact gathers
    cm = torch.from_numpy(z["transform_logmel23_cummn"]).to(dev)
    assert np.array_equal(feature.splice_subsample(cm, ctx, sub).cpu().numpy(), z["subsample_Y"])
