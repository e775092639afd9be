"""Utterance sharding for multi-GPU inference (one process per GPU, no data-path collective).

Utterances / streams are independent through the whole forward, so rank r of W simply owns a
contiguous slice of the work list; the only communication is the throughput bookkeeping of a
benchmark or an evaluation (sum of frames, max of wall time) -- a scalar all-reduce, RCCL on
GPUs (backend "nccl"), gloo in the CPU tests."""
from typing import Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Balanced contiguous split: the first n_items % world ranks get one extra item."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def job_throughput(frames_local: float, seconds_local: float, device=None) -> float:
    """Whole-job frames/s = (sum over ranks of frames) / (max over ranks of seconds)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return frames_local / seconds_local
    f = torch.tensor([frames_local], dtype=torch.float64, device=device)
    t = torch.tensor([seconds_local], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(f.item() / t.item())
