"""Data-side helpers the multi-GPU training loop needs; host-side index arithmetic only.

    MyDistributedSampler     LS-EEND/data_loaders/utils/my_distributed_sampler.py:19-95 -- yields (index, seed) pairs so that
                             on-the-fly chunking is reproducible on any machine; rank r takes every num_replicas-th pair
    count_frames / gen_frame_indices / chunk_table
                             {FS,LS}-EEND/datasets/diarization_dataset*.py:12-29,69-81 -- fixed chunk grid over recordings
    on_the_fly_chunk         LS-EEND/datasets/diarization_dataset_on_the_fly.py:87-104 -- random chunk start drawn from a
                             PCG64 stream seeded with the sampler's per-item seed
    select_epoch_checkpoints FS-EEND/train_dia.py:170 -- which `epoch=N-...ckpt` files enter the average
(the averaging itself: trainer.average_checkpoints)
"""
import math
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


class MyDistributedSampler:
    """(index, seed) sampler with DistributedSampler's sharding arithmetic.  `num_replicas` / `rank` default to the
    initialised process group, or (1, 0) without one (the reference's single-GPU fallback, :33-38)."""

    def __init__(self, dataset, num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True,
                 seed: int = 0, drop_last: bool = False):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
                rank = dist.get_rank() if rank is None else rank
            else:
                num_replicas, rank = 1, 0
        if rank >= num_replicas or rank < 0:
            raise ValueError(f"invalid rank {rank} for {num_replicas} replicas")
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.shuffle, self.seed, self.drop_last, self.epoch = shuffle, seed, drop_last, 0
        n = len(dataset)
        if drop_last and n % num_replicas != 0:
            self.num_samples = math.ceil((n - num_replicas) / num_replicas)
        else:
            self.num_samples = math.ceil(n / num_replicas)
        self.total_size = self.num_samples * num_replicas
        self.last_epoch = -1

    def __iter__(self) -> Iterator[Tuple[int, int]]:
        n = len(self.dataset)
        g = torch.Generator()
        if self.shuffle:
            g.manual_seed(self.seed + self.epoch)
            if self.last_epoch < self.epoch:
                self.last_epoch = self.epoch
            indices = torch.randperm(n, generator=g).tolist()
        else:
            g.manual_seed(self.seed)
            indices = list(range(n))
        seeds = [torch.randint(high=9999999999, size=(1,), generator=g)[0].item() for _ in range(n)]
        pairs = list(zip(indices, seeds))
        if not self.drop_last:
            pad = self.total_size - len(pairs)
            if pad <= len(pairs):
                pairs += pairs[:pad]
            else:
                pairs += (pairs * math.ceil(pad / len(pairs)))[:pad]
        else:
            pairs = pairs[:self.total_size]
        pairs = pairs[self.rank:self.total_size:self.num_replicas]
        assert len(pairs) == self.num_samples
        return iter(pairs)

    def __len__(self) -> int:
        return self.num_samples

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch


def count_frames(data_len: int, size: int, step: int) -> int:
    """no padding at the edges, the last remaining samples are ignored (diarization_dataset.py:12-14)"""
    return int((data_len - size + step) / step)


def gen_frame_indices(data_length: int, size: int = 2000, step: int = 2000, use_last_samples: bool = False,
                      label_delay: int = 0, subsampling: int = 1):
    """diarization_dataset.py:17-29: the chunk grid of one recording (+ the shorter tail chunk if asked for)."""
    i = -1
    for i in range(count_frames(data_length, size, step)):
        yield i * step, i * step + size
    if use_last_samples and i * step + size < data_length:
        if data_length - (i + 1) * step - label_delay > 0:
            yield (i + 1) * step, data_length
    elif i == -1:
        yield 0, data_length


def chunk_table(recordings: Sequence[Tuple[str, float]], chunk_size: int, chunk_step: int, frame_shift: int, rate: int,
                subsampling: int = 1, use_last_samples: bool = False, label_delay: int = 0, on_the_fly: bool = False):
    """(rec, [data_len,] start_frame, end_frame) rows, frames BEFORE subsampling.
    FS-EEND/datasets/diarization_dataset.py:69-81; with on_the_fly the LS variant that also records the length (:77-81)."""
    rows = []
    for rec, dur in recordings:
        data_len = int(dur * rate / frame_shift)
        data_len = int(data_len / subsampling)
        for st, ed in gen_frame_indices(data_len, chunk_size, chunk_step, use_last_samples, label_delay=label_delay,
                                        subsampling=subsampling):
            if on_the_fly:
                rows.append((rec, data_len * subsampling, st * subsampling, ed * subsampling))
            else:
                rows.append((rec, st * subsampling, ed * subsampling))
    return rows


def on_the_fly_chunk(row, seed: int, chunk_size: int, subsampling: int, data_type: str = "train"):
    """diarization_dataset_on_the_fly.py:87-104: for training the chunk start is drawn uniformly from the recording
    (rng.choice over range(data_len), PCG64(seed)); evaluation keeps the grid's (st, ed)."""
    rec, data_len, st, ed = row
    if data_type == "train":
        rng = np.random.default_rng(np.random.PCG64(seed))
        st = int(rng.choice(range(data_len)))
        ed = min(st + chunk_size * subsampling, data_len)
    return rec, st, ed


def select_epoch_checkpoints(files: Sequence[str], start_epoch: int, end_epoch: int) -> List[str]:
    """FS-EEND/train_dia.py:170: Lightning's `epoch=N-step=M.ckpt` files with start <= N <= end."""
    out = []
    for x in files:
        if ".ckpt" in x and "epoch" in x:
            n = int(x.split("=")[1].split("-")[0])
            if start_epoch <= n <= end_epoch:
                out.append(x)
    return out
