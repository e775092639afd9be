"""CPU: the feature front-end restatement (oracle/feature_ref.py).  librosa is absent (parity unpinned for the
two librosa calls); the STFT is cross-checked against torch.stft, which implements the same centred, padded-
window definition, and the filterbank against the structural properties of the Slaney definition."""
import numpy as np
import pytest
import torch

from oracle import feature_ref as F


@pytest.mark.parametrize("n,mode", [(8000, "constant"), (8123, "constant"), (800, "reflect"), (79, "constant"), (4001, "reflect")])
def test_stft_matches_torch_stft(n, mode):
    g = np.random.default_rng(n)
    y = (g.standard_normal(n) * 0.1).astype(np.float32)
    got = F.stft(y, 200, 80, pad_mode=mode)
    ref = torch.stft(torch.from_numpy(y), n_fft=256, hop_length=80, win_length=200, window=torch.hann_window(200, periodic=True),
                     center=True, pad_mode=mode, return_complex=True).t().numpy()
    if n % 80 == 0:
        ref = ref[:-1]
    assert got.shape == ref.shape == ((n // 80 if n % 80 == 0 else 1 + n // 80), 129)
    assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_mel_filterbank_slaney_properties():
    M = F.mel_filterbank(8000, 256, 23)
    assert M.shape == (23, 129) and M.dtype == np.float32 and (M >= 0).all()
    freqs = np.linspace(0, 4000, 129)
    centres = (M * freqs).sum(1) / M.sum(1)
    assert (np.diff(centres) > 0).all()                      # triangles ordered in frequency
    assert (np.diff(M.argmax(1)) >= 0).all()
    # Slaney area normalisation: each triangle integrates to ~1 in Hz (bin width 4000/128)
    area = M.sum(1) * (4000 / 128)
    assert np.abs(area - 1).max() < 0.12
    # mel scale is linear (200/3 Hz per mel) below 1 kHz, logarithmic above
    assert abs(F._hz_to_mel(1000.0) - 15.0) < 1e-12 and abs(F._mel_to_hz(F._hz_to_mel(3000.0)) - 3000.0) < 1e-9
    assert M[0, 0] == 0 and M[-1, -1] == 0                  # first triangle starts at 0 Hz, last ends at Nyquist


def test_transform_splice_subsample_follow_the_reference_lines():
    g = np.random.default_rng(1)
    y = (g.standard_normal(16000) * 0.05).astype(np.float32)
    Y = F.stft(y)
    lm = F.transform(Y, "logmel23")
    assert lm.shape == (200, 23) and lm.dtype == np.float32
    P = (np.abs(Y) ** 2) @ F.mel_filterbank().T
    assert np.allclose(lm, np.log10(np.maximum(P, 1e-10)), atol=1e-6)
    cm = F.transform(Y, "logmel23_cummn")
    assert np.allclose(cm, lm - np.cumsum(lm, 0) / np.arange(1, 201)[:, None], atol=1e-5)
    assert np.allclose(F.transform(Y, "logmel23_mn"), lm - lm.mean(0), atol=1e-5)
    S = F.splice(lm, 7)
    assert S.shape == (200, 345)
    assert np.array_equal(S[0, :7 * 23], np.zeros(161, np.float32)) and np.array_equal(S[0, 7 * 23:8 * 23], lm[0])
    assert np.array_equal(S[100].reshape(15, 23), lm[93:108]) and np.array_equal(S[199, 8 * 23:], np.zeros(161, np.float32))
    out = F.extract_fbank_wave(y)
    assert out.shape == (20, 345) and np.array_equal(out, S[::10])
    with pytest.raises(ValueError):
        F.transform(Y, "logmel23_swn")


def test_silence_hits_the_log_floor():
    out = F.extract_fbank_wave(np.zeros(1600, np.float32))
    assert out.shape == (2, 345)
    assert np.all((out == 0) | (np.abs(out + 10.0) < 2e-6))           # log10(float32(1e-10)) or splice padding


def test_unpinned_mel_table_matches_an_independent_librosa_compatible_implementation():
    """librosa itself is absent (the oracle stays "parity unpinned" for `librosa.filters.mel`); transformers.audio_utils
    ships an independently written Slaney filterbank that its own test-suite holds against librosa -- a second source."""
    au = pytest.importorskip("transformers.audio_utils")
    M2 = au.mel_filter_bank(num_frequency_bins=129, num_mel_filters=23, min_frequency=0.0, max_frequency=4000.0, sampling_rate=8000,
                            norm="slaney", mel_scale="slaney").T
    M = F.mel_filterbank(8000, 256, 23)
    assert M2.shape == M.shape and np.abs(M - M2).max() < 1e-8


@pytest.mark.parametrize("n", [8123, 8000, 1601])
def test_unpinned_stft_power_matches_scipy_signal(n):
    """Second cross-check of the `librosa.stft` restatement: scipy.signal.stft frames the same samples (boundary padding of
    win_length/2 = librosa's centred window inside its n_fft/2 padding) and zero-pads each segment at the END, i.e. the
    same magnitudes with a different linear phase; its 'spectrum' scaling divides by sum(window)."""
    import scipy.signal as ss
    g = np.random.default_rng(n)
    y = (g.standard_normal(n) * 0.1).astype(np.float32)
    Y = F.stft(y, 200, 80)
    w = ss.get_window("hann", 200, fftbins=True)
    _, _, Z = ss.stft(y, fs=8000, window=w, nperseg=200, noverlap=120, nfft=256, boundary="zeros", padded=False)
    Z = Z.T * w.sum()
    assert Z.shape[0] >= Y.shape[0]
    assert np.abs(np.abs(Y) - np.abs(Z[:Y.shape[0]])).max() < 2e-5 * max(1.0, np.abs(Y).max())
