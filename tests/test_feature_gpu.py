"""GPU: the HIP feature front-end against the oracle restatement (oracle/feature_ref.py; parity of the two
librosa-defined pieces is unpinned, see its header).  Tolerance: 2e-4 absolute on log10-mel values (fp32 DFT by
matrix product vs float64 FFT; the log amplifies relative error near the 1e-10 floor only for exact silence,
which is tested separately)."""
import numpy as np
import pytest
import torch

from oracle import feature_ref as R

pytestmark = pytest.mark.gpu
TOL = 2e-4


def wave(n, seed, kind="speechlike"):
    g = np.random.default_rng(seed)
    t = np.arange(n) / 8000.0
    y = 0.05 * g.standard_normal(n)
    if kind == "speechlike":
        env = 0.5 * (1 + np.sin(2 * np.pi * 0.7 * t + 1.0)) * (g.random(n) > 0.02)
        y = y * 0.2 + env * (0.3 * np.sin(2 * np.pi * 180 * t) + 0.2 * np.sin(2 * np.pi * 1230 * t + 0.3) + 0.1 * np.sin(2 * np.pi * 3100 * t))
    return y.astype(np.float32)


@pytest.mark.parametrize("n,tr,mode", [(8000, "logmel23", "constant"), (16123, "logmel23_cummn", "constant"), (4001, "logmel23_mn", "constant"),
                                        (799, "logmel23", "reflect"), (80, "logmel23", "constant"), (79, "logmel23_cummn", "constant"),
                                        (160000, "logmel23_cummn", "constant"), (50000, "logmel23", "reflect")])
def test_logmel_and_fbank_vs_oracle(hip_lib, dev, n, tr, mode):
    from fs_eend_amd import feature
    y = wave(n, n)
    want_lm = R.transform(R.stft(y, 200, 80, pad_mode=mode), tr)
    got_lm = feature.logmel(torch.from_numpy(y).to(dev), input_transform=tr, pad_mode=mode).cpu().numpy()
    assert got_lm.shape == want_lm.shape
    assert np.abs(got_lm - want_lm).max() < TOL, np.abs(got_lm - want_lm).max()
    want = R.extract_fbank_wave(y, input_transform=tr, pad_mode=mode)
    got = feature.extract_fbank_wave(torch.from_numpy(y).to(dev), input_transform=tr, pad_mode=mode).cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() < TOL


def test_splice_subsample_exact(hip_lib, dev):
    from fs_eend_amd import feature
    g = torch.Generator().manual_seed(0)
    for T, F, c, s in [(200, 23, 7, 10), (1, 23, 7, 10), (37, 5, 0, 1), (1001, 23, 7, 10), (15, 3, 2, 4)]:
        Y = torch.randn(T, F, generator=g)
        got = feature.splice_subsample(Y.to(dev), c, s).cpu().numpy()
        assert np.array_equal(got, R.splice(Y.numpy(), c)[::s])


def test_silence_and_long_cumulative_mean(hip_lib, dev):
    from fs_eend_amd import feature
    out = feature.extract_fbank_wave(torch.zeros(1600, device=dev), input_transform="logmel23").cpu().numpy()
    assert out.shape == (2, 345) and np.all((out == 0) | (np.abs(out + 10.0) < 2e-6))
    # ten minutes: the running mean is carried in fp64 on the device, numpy's float32 cumsum drifts by ~1e-6
    y = wave(4_800_000, 3)
    want = R.transform(R.stft(y), "logmel23_cummn")
    got = feature.logmel(torch.from_numpy(y).to(dev), input_transform="logmel23_cummn").cpu().numpy()
    assert got.shape == want.shape == (60000, 23)
    assert np.abs(got - want).max() < 5e-4


def test_features_feed_the_model(hip_lib, dev):
    """wave -> features -> FS-EEND model.test -> RTTM, all on the device (dia_pred.predict, FS-EEND/dia_pred.py:22-63)."""
    from oracle import fixtures as FX
    from fs_eend_amd import feature, postproc
    from tests.helpers import build_fs_mirror
    meta, _ = FX.load_case("fs_full_T500_c6")
    m = build_fs_mirror(meta).to(dev)
    feat = feature.extract_fbank_wave(torch.from_numpy(wave(8000 * 30, 9)).to(dev), input_transform="logmel23")
    assert feat.shape == (300, 345)
    preds, _, _ = m.test([feat], [feat.shape[0]], max_nspks=6)
    rttm = postproc.make_rttm("utt", torch.sigmoid(preds[0][:, 1:]))
    assert torch.isfinite(preds[0]).all() and isinstance(rttm, dict)
