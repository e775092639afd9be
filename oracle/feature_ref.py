"""CPU restatement of the reference's feature front-end -- TEST INFRASTRUCTURE
ONLY: imported by tests/, never by the product path.

    stft / transform('logmel23' | 'logmel23_mn' | 'logmel23_cummn') / splice / subsample / extract_fbank
        LS-EEND/datasets/feature.py:43-90,133-191,324-336   (FS-EEND/datasets/feature.py:26-161,356-368)

PARITY UNPINNED for the two librosa calls: the reference computes the STFT with `librosa.stft` and the
filterbank with `librosa.filters.mel`; librosa (requirements.txt: "librosa", no version pin) is absent from
this image and the reference ships no feature fixtures, so both are restated from librosa's published
definition:
  * stft(y, n_fft=256, hop_length=80, win_length=200): center=True pads n_fft//2 samples on both sides
    (pad_mode="constant" since librosa 0.10, "reflect" before: a parameter here, default "constant");
    window = scipy.signal.get_window("hann", 200, fftbins=True), zero-padded to n_fft on both sides
    (28 + 200 + 28); frame t = padded[t*hop : t*hop + n_fft]; rfft -> 129 bins; 1 + len(y)//hop frames.
  * filters.mel(sr=8000, n_fft=256, n_mels=23): Slaney mel scale (linear below 1 kHz, log above, 200/3 Hz
    per mel), fmin=0, fmax=sr/2, triangular weights, Slaney area normalisation 2/(f[i+2]-f[i]), float32.
The STFT restatement is cross-checked against torch.stft (built to match librosa) and, magnitudes only, against
scipy.signal.stft; the filterbank equals transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney") -- an
independently written table that project holds against librosa -- to 1e-9 (tests/test_oracle_feature.py, "unpinned" tests).
That is two independent sources agreeing, NOT an output of librosa itself: the status stays "parity unpinned".
everything downstream of the two librosa calls (|.|^2 . mel^T, log10, the mean normalisations, splice,
subsample, the "drop the last frame when len % hop == 0" rule) follows the reference source line by line.
"""
import numpy as np


def hann_padded(win_length=200, n_fft=256):
    n = np.arange(win_length)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)          # periodic ("fftbins=True") Hann
    lpad = (n_fft - win_length) // 2
    return np.pad(w, (lpad, n_fft - win_length - lpad))


def stft(data, frame_size=200, frame_shift=80, pad_mode="constant"):
    """feature.py:166-191: (n_frames, n_bins) complex64; the excess last frame is dropped when len % shift == 0."""
    data = np.asarray(data, dtype=np.float32)
    n_fft = 1 << (frame_size - 1).bit_length()
    w = hann_padded(frame_size, n_fft).astype(np.float32)
    yp = np.pad(data, n_fft // 2, mode=pad_mode)
    n_frames = 1 + len(data) // frame_shift
    idx = np.arange(n_fft)[None, :] + frame_shift * np.arange(n_frames)[:, None]
    Y = np.fft.rfft(yp[idx] * w[None, :], axis=1).astype(np.complex64)
    return Y[:-1] if len(data) % frame_shift == 0 else Y


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=8000, n_fft=256, n_mels=23):
    """librosa.filters.mel(sr, n_fft, n_mels) with its defaults (fmin=0, fmax=sr/2, htk=False, norm='slaney')."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


def transform(Y, transform_type, dtype=np.float32):
    """feature.py:43-131 for the log-mel-23 family used by the shipped configs."""
    Y = np.abs(Y)
    if transform_type not in ("logmel23", "logmel23_mn", "logmel23_cummn"):
        raise ValueError("Unknown transform_type: %s" % transform_type)
    n_fft = 2 * (Y.shape[1] - 1)
    mel_basis = mel_filterbank(8000, n_fft, 23)
    Y = np.dot(Y ** 2, mel_basis.T)
    Y = np.log10(np.maximum(Y, 1e-10))
    if transform_type == "logmel23_mn":
        Y = Y - np.mean(Y, axis=0)
    elif transform_type == "logmel23_cummn":
        cumsum = np.cumsum(Y, axis=0)
        idx = np.arange(1, Y.shape[0] + 1)
        Y = Y - cumsum / idx[:, None]
    return Y.astype(dtype)


def splice(Y, context_size=0):
    """feature.py:141-163: row t = concat(Y[t-c] .. Y[t+c]) with zero rows outside."""
    Yp = np.pad(Y, [(context_size, context_size), (0, 0)], "constant")
    return np.stack([Yp[t:t + 2 * context_size + 1].reshape(-1) for t in range(Y.shape[0])]) if Y.shape[0] else \
        np.zeros((0, Y.shape[1] * (2 * context_size + 1)), Y.dtype)


def extract_fbank_wave(data, context_size=7, frame_size=200, frame_shift=80, input_transform="logmel23", subsampling=10,
                       pad_mode="constant"):
    """feature.py:324-336 from the decoded waveform on (soundfile is absent here)."""
    Y = transform(stft(data, frame_size, frame_shift, pad_mode), input_transform)
    return splice(Y, context_size)[::subsampling].astype(np.float32)
