"""CPU restatement of the reference's output-side post-processing --
TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

  * make_rttm             FS-EEND/train/utils/make_rttm.py:10-28 (LS-EEND copy is identical)
  * calc_diarization_error / report_diarization_error
                          FS-EEND/train/utils/loss.py:198-254 (LS-EEND train/utils/loss.py:215-275)

Pinned: tests/golden/post_*.npz hold the outputs of the reference's own function bodies (executed from
/root/reference by oracle/gen_golden_post.py; the modules themselves need h5py / torchmetrics, which this
image lacks, so the generator evaluates just those function definitions) on seeded inputs.
Integer / index work throughout: parity is bit exact.
"""
from collections import defaultdict
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import Tensor


def median_binary(x: np.ndarray, k: int) -> np.ndarray:
    """scipy.signal.medfilt(x, (k, 1)) on a (T, S) 0/1 array: zero padding, so the median of the k values
    is 1 exactly when at least k//2 + 1 of them are 1."""
    if k <= 1:
        return x.astype(np.int64)
    T = x.shape[0]
    pad = np.zeros((k // 2, x.shape[1]), dtype=np.int64)
    xp = np.concatenate([pad, x.astype(np.int64), pad], axis=0)
    cs = np.concatenate([np.zeros((1, x.shape[1]), dtype=np.int64), np.cumsum(xp, axis=0)], axis=0)
    win = cs[k:k + T] - cs[:T]
    return (win >= k // 2 + 1).astype(np.int64)


def activity(pred: Tensor, threshold: float = 0.5, median: int = 11) -> np.ndarray:
    """make_rttm.py:12-15: pred (T, S) probabilities -> 0/1 (T, S) after threshold and median filter."""
    dec = (pred > threshold).cpu().numpy().astype(np.int64)
    return median_binary(dec, median) if median > 1 else dec


def segments(act: np.ndarray) -> List[List[Tuple[int, int]]]:
    """make_rttm.py:18-21: per speaker, (start, end) frame pairs where the zero-padded track changes."""
    out = []
    for s in range(act.shape[1]):
        f = np.concatenate([[0], act[:, s], [0]])
        ch = np.nonzero(np.diff(f) != 0)[0]
        out.append([(int(a), int(b)) for a, b in zip(ch[::2], ch[1::2])])
    return out


def make_rttm(rec: str, pred: Tensor, frame_shift=80, threshold=0.5, median=11, subsampling=10, sampling_rate=8000):
    rttm = defaultdict(list)
    fmt = "SPEAKER {:s} 1 {:7.2f} {:7.2f} <NA> <NA> {:s} <NA>"
    for spkid, segs in enumerate(segments(activity(pred, threshold, median))):
        for s, e in segs:
            # the reference formats 0-dim torch tensors: int64 * int / int -> float32
            st = torch.tensor(s) * frame_shift * subsampling / sampling_rate
            du = (torch.tensor(e) - torch.tensor(s)) * frame_shift * subsampling / sampling_rate
            rttm[str(spkid)].append(fmt.format(rec, st, du, rec + "_" + str(spkid)))
    return rttm


DER_KEYS = ("speech_scored", "speech_miss", "speech_falarm", "speaker_scored", "speaker_miss", "speaker_falarm",
            "speaker_error", "correct", "diarization_error", "frames")


def calc_diarization_error(pred: Tensor, label: Tensor, label_delay: int = 0) -> Dict[str, float]:
    """loss.py:198-236: frame-level DER counters of (T, C) pre-activations against (T, C) 0/1 labels."""
    label = label[: len(label) - label_delay]
    dec = torch.sigmoid(pred[label_delay:]) > 0.5
    n_ref = label.sum(dim=-1).long()
    n_sys = dec.sum(dim=-1).long()
    res = {}
    res["speech_scored"] = int((n_ref > 0).sum())
    res["speech_miss"] = int(((n_ref > 0) & (n_sys == 0)).sum())
    res["speech_falarm"] = int(((n_ref == 0) & (n_sys > 0)).sum())
    res["speaker_scored"] = int(n_ref.sum())
    res["speaker_miss"] = int(torch.clamp(n_ref - n_sys, min=0).sum())
    res["speaker_falarm"] = int(torch.clamp(n_sys - n_ref, min=0).sum())
    n_map = ((label == 1) & (dec == 1)).sum(dim=-1)
    res["speaker_error"] = int((torch.min(n_ref, n_sys) - n_map).sum())
    res["correct"] = float((label == dec).sum() / label.shape[1])
    res["diarization_error"] = res["speaker_miss"] + res["speaker_falarm"] + res["speaker_error"]
    res["frames"] = len(label)
    return res


def report_diarization_error(ys, labels, label_delay: int = 0):
    """loss.py:239-254."""
    stats = defaultdict(list)
    for y, t in zip(ys, labels):
        for k, v in calc_diarization_error(y, t, label_delay).items():
            stats[k].append(float(v))
    return stats
