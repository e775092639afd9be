"""Feature front-end on the GPU (csrc/feature.hip; no CPU fallback): the reference's
`datasets/feature.py` chain  stft -> transform('logmel23' | 'logmel23_mn' | 'logmel23_cummn') -> splice ->
subsample  ({LS,FS}-EEND/datasets/feature.py: stft :166-191, transform :43-131, splice :141-163,
subsample :133-138, extract_fbank :324-336) fused into three launches.

The two librosa calls of the reference (librosa.stft, librosa.filters.mel; librosa is not installable here) are
replaced by tables built below from librosa's published definitions: periodic Hann window of `frame_size`
samples centred in an n_fft frame, centred framing with n_fft // 2 samples of padding (pad_mode "constant" as in
librosa >= 0.10, or "reflect" as in older releases), Slaney mel scale and area normalisation.
"""
import math

import torch

from . import lib as _lib

_F32 = torch.float32
_TABLES = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _hz_to_mel(f):
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    return min_log_hz / f_sp + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp


def _mel_to_hz(m):
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return min_log_hz * math.exp(logstep * (m - min_log_mel)) if m >= min_log_mel else f_sp * m


def mel_filterbank(sr=8000, n_fft=256, n_mels=23):
    """(n_mels, 1 + n_fft/2) float64 tensor: triangular Slaney filters, fmin = 0, fmax = sr / 2, area-normalised."""
    nb = 1 + n_fft // 2
    fftfreqs = [sr / 2.0 * i / (nb - 1) for i in range(nb)]
    lo, hi = _hz_to_mel(0.0), _hz_to_mel(sr / 2.0)
    mel_f = [_mel_to_hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    W = torch.zeros(n_mels, nb, dtype=torch.float64)
    for i in range(n_mels):
        enorm = 2.0 / (mel_f[i + 2] - mel_f[i])
        for b, fr in enumerate(fftfreqs):
            lower = (fr - mel_f[i]) / (mel_f[i + 1] - mel_f[i])
            upper = (mel_f[i + 2] - fr) / (mel_f[i + 2] - mel_f[i + 1])
            W[i, b] = max(0.0, min(lower, upper)) * enorm
    return W


def _tables(dev):
    key = str(dev)
    if key not in _TABLES:
        n_fft, win, lpad = 256, 200, 28
        k = torch.arange(win, dtype=torch.float64)
        w = 0.5 - 0.5 * torch.cos(2.0 * math.pi * k / win)                       # periodic Hann
        n = torch.arange(129, dtype=torch.float64)
        ang = 2.0 * math.pi * torch.outer(k + lpad, n) / n_fft
        dft = torch.zeros(win, 288, dtype=torch.float64)
        dft[:, :129] = w[:, None] * torch.cos(ang)
        dft[:, 144:273] = -w[:, None] * torch.sin(ang)
        melT = torch.zeros(132, 32, dtype=torch.float64)
        melT[:129, :23] = mel_filterbank(8000, n_fft, 23).to(torch.float32).to(torch.float64).t()   # librosa returns float32
        _TABLES[key] = (dft.to(_F32).to(dev).contiguous(), melT.to(_F32).to(dev).contiguous())
    return _TABLES[key]


_MODES = {"logmel23": 0, "logmel23_mn": 1, "logmel23_cummn": 2}


def logmel(data, frame_size=200, frame_shift=80, input_transform="logmel23", pad_mode="constant"):
    """stft + transform of the reference for the log-mel-23 family: 1-D float32 GPU waveform -> (n_frames, 23)."""
    if not isinstance(data, torch.Tensor) or not data.is_cuda:
        raise _lib.EendHipError("feature.logmel: expected a GPU tensor (the HIP path has no CPU fallback)")
    if frame_size != 200 or frame_shift != 80:
        raise NotImplementedError("HIP feature front-end is specialised for frame_size=200, frame_shift=80 (all shipped configs)")
    if input_transform not in _MODES:
        raise ValueError("Unknown transform_type: %s" % input_transform)
    L = _lib.load()
    data = data.to(_F32).contiguous().view(-1)
    n = data.numel()
    n_frames = 1 + n // frame_shift
    if n % frame_shift == 0:
        n_frames -= 1                     # feature.py:184-188: the excess last frame is dropped
    dft, melT = _tables(data.device)
    if n_frames <= 0:
        return torch.zeros(0, 23, dtype=_F32, device=data.device)
    first = -100                          # lpad - n_fft // 2
    if pad_mode == "reflect":
        data = torch.nn.functional.pad(data.view(1, 1, -1), (128, 128), mode="reflect").view(-1).contiguous()
        first = 28
    elif pad_mode != "constant":
        raise ValueError("pad_mode must be 'constant' or 'reflect'")
    Y = torch.empty(n_frames, 23, dtype=_F32, device=data.device)
    _lib.check(L.eend_stft_logmel23_f32(data.data_ptr(), data.numel(), first, n_frames, dft.data_ptr(), melT.data_ptr(),
                                        Y.data_ptr(), _stream()), "eend_stft_logmel23_f32")
    mode = _MODES[input_transform]
    if mode:
        out = torch.empty_like(Y)
        _lib.check(L.eend_feature_meannorm_f32(Y.data_ptr(), out.data_ptr(), n_frames, 23, mode, _stream()), "eend_feature_meannorm_f32")
        Y = out
    return Y


def splice_subsample(Y, context_size=0, subsampling=1):
    """splice(Y, context_size)[::subsampling] of the reference in one launch; Y (T, F) float32 GPU tensor."""
    L = _lib.load()
    if not Y.is_cuda:
        raise _lib.EendHipError("feature.splice_subsample: expected a GPU tensor")
    Y = Y.to(_F32).contiguous()
    T, F = Y.shape
    To = (T + subsampling - 1) // subsampling
    out = torch.empty(To, F * (2 * context_size + 1), dtype=_F32, device=Y.device)
    if T:
        _lib.check(L.eend_splice_subsample_f32(Y.data_ptr(), T, F, context_size, subsampling, out.data_ptr(), _stream()),
                   "eend_splice_subsample_f32")
    return out


def splice(Y, context_size=0):
    return splice_subsample(Y, context_size, 1)


def subsample(Y, T, subsampling=1):
    return Y[::subsampling], T[::subsampling]


def extract_fbank_wave(data, context_size=7, frame_size=200, frame_shift=80, input_transform=None, subsampling=10,
                       pad_mode="constant"):
    """extract_fbank (feature.py:324-336) from the decoded waveform on: (ceil(n_frames / subsampling), 23 * 15)."""
    if not input_transform:
        raise NotImplementedError("only the log-mel-23 transforms are implemented on the GPU")
    return splice_subsample(logmel(data, frame_size, frame_shift, input_transform, pad_mode), context_size, subsampling)


def extract_fbank(wav_path, context_size=7, frame_size=200, frame_shift=80, input_transform=None, subsampling=10):
    """Same signature as the reference; decoding the file needs `soundfile`, which this image does not have."""
    try:
        import soundfile as sf
    except ImportError as e:
        raise ImportError("extract_fbank(wav_path) needs the `soundfile` package to decode audio; "
                          "use extract_fbank_wave(waveform_tensor, ...) with an already decoded signal") from e
    data, _rate = sf.read(wav_path, dtype="float32")
    dev = torch.device("cuda", torch.cuda.current_device())
    return extract_fbank_wave(torch.from_numpy(data).to(dev), context_size, frame_size, frame_shift, input_transform, subsampling)
