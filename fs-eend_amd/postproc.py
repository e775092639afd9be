"""Host-side mirror of the reference's output post-processing, running on the HIP kernels of
csrc/postproc.hip (no CPU fallback: tensors must live on the GPU).

    make_rttm                  FS-EEND/train/utils/make_rttm.py:10-28 (LS-EEND copy identical)
    calc_diarization_error     FS-EEND/train/utils/loss.py:198-236   (LS-EEND train/utils/loss.py:215-257)
    report_diarization_error   FS-EEND/train/utils/loss.py:239-254

Same names, arguments and return structure as the reference, so `dia_pred.py` / `streaming_infer_dia.py` /
the Lightning validation step can import them from the drop-in `train.utils` shims.
"""
from collections import defaultdict

import torch

from . import lib as _lib

_U8, _I32, _F32 = torch.uint8, torch.int32, torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _gpu_f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.EendHipError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != _F32:
        t = t.to(_F32)
    if t.dim() != 2 or t.stride(1) != 1:
        t = t.contiguous()
    return t


def _to_device(t, name):
    """Drop-in entry points accept what the reference's drivers pass -- CPU tensors (`pred.detach().cpu()`,
    FS streaming_infer_dia.py:95-99, train/oln_tfm_enc_dec.py:262) -- and move them to the current GPU; with no GPU
    there is nothing to run on, and that is an error (no CPU fallback)."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if t.is_cuda:
        return t
    if not torch.cuda.is_available():
        raise _lib.EendHipError(f"{name}: no GPU available (the HIP path has no CPU fallback)")
    return t.to(torch.device("cuda", torch.cuda.current_device()))


def activity(pred, threshold=0.5, median=11):
    """(T, S) probabilities -> uint8 (T, S): threshold, then median filter along time (make_rttm.py:12-15)."""
    L = _lib.load()
    pred = _gpu_f32(pred, "pred")
    T, S = pred.shape
    out = torch.empty(T, S, dtype=_U8, device=pred.device)
    _lib.check(L.eend_activity_median_u8(pred.data_ptr(), pred.stride(0), T, S, float(threshold), int(median) if median > 1 else 1,
                                         out.data_ptr(), _stream()), "eend_activity_median_u8")
    return out


def segments(act):
    """uint8 (T, S) -> list over speakers of [(start_frame, end_frame), ...] (make_rttm.py:18-21)."""
    L = _lib.load()
    T, S = act.shape
    cap = T + 2                                   # a track of T frames has at most T + 1 change points
    changes = torch.empty(S, cap, dtype=_I32, device=act.device)
    counts = torch.empty(S, dtype=_I32, device=act.device)
    _lib.check(L.eend_activity_segments_i32(act.data_ptr(), T, S, changes.data_ptr(), counts.data_ptr(), cap, _stream()),
               "eend_activity_segments_i32")
    n = counts.cpu().tolist()
    ch = changes[:, : max(n) if n else 0].cpu().tolist() if n and max(n) > 0 else [[] for _ in range(S)]
    return [list(zip(ch[s][0:n[s]:2], ch[s][1:n[s]:2])) for s in range(S)]


def make_rttm(rec, pred, frame_shift=80, threshold=0.5, median=11, subsampling=10, sampling_rate=8000):
    pred = _to_device(pred, "pred")
    rttm = defaultdict(list)
    fmt = "SPEAKER {:s} 1 {:7.2f} {:7.2f} <NA> <NA> {:s} <NA>"
    for spkid, segs in enumerate(segments(activity(pred, threshold, median))):
        if not segs:
            continue
        # the reference formats 0-dim torch tensors computed as int64 * int / int -> float32; the same
        # arithmetic on the whole vector of change points gives the same float32 values
        se = torch.tensor(segs, dtype=torch.int64)
        st = (se[:, 0] * frame_shift * subsampling / sampling_rate).tolist()
        du = ((se[:, 1] - se[:, 0]) * frame_shift * subsampling / sampling_rate).tolist()
        name = rec + "_" + str(spkid)
        rttm[str(spkid)] = [fmt.format(rec, a, b, name) for a, b in zip(st, du)]
    return rttm


def der_counters(pred, label, label_delay=0):
    """Device-side counters (uint64 tensor of 8, still on the GPU; no host sync)."""
    L = _lib.load()
    pred, label = _gpu_f32(pred, "pred"), _gpu_f32(label, "label")
    if pred.shape != label.shape:
        raise _lib.EendHipError("calc_diarization_error: pred and label shapes differ")
    T, C = pred.shape
    out = torch.empty(8, dtype=torch.int64, device=pred.device)
    _lib.check(L.eend_der_counters_u64(pred.data_ptr(), pred.stride(0), label.data_ptr(), label.stride(0), T, C, int(label_delay),
                                       out.data_ptr(), _stream()), "eend_der_counters_u64")
    return out


def calc_diarization_error(pred, label, label_delay=0):
    pred, label = _to_device(pred, "pred"), _to_device(label, "label")
    T, C = pred.shape
    v = der_counters(pred, label, label_delay).cpu().tolist()
    res = {}
    (res["speech_scored"], res["speech_miss"], res["speech_falarm"], res["speaker_scored"], res["speaker_miss"],
     res["speaker_falarm"], res["speaker_error"]) = v[:7]
    # the reference divides two tensors: int64 / python int -> float32
    res["correct"] = float(torch.tensor(v[7]) / C)
    res["diarization_error"] = res["speaker_miss"] + res["speaker_falarm"] + res["speaker_error"]
    res["frames"] = T - label_delay
    if res["speaker_error"] < 0:
        raise Exception("spk error")
    return res


def report_diarization_error(ys, labels, label_delay=0):
    stats_batch = defaultdict(list)
    for y, t in zip(ys, labels):
        for k, v in calc_diarization_error(y, t, label_delay).items():
            stats_batch[k].append(float(v))
    return stats_batch
