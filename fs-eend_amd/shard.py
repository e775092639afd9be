"""Multi-GPU plumbing: utterance sharding for inference (one process per GPU, no data-path collective) and the one
exchange step of data-parallel training (mean all-reduce of the flat gradient buffer).

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


def flat_layout(named_shapes, align: int = 4):
    """Offsets of every parameter in the flat f32 buffers of the training step (named_parameters order, each slice
    aligned to `align` elements = 16 bytes).  Returns ({name: offset}, total elements).  Every rank derives the same
    layout from the same model, so ONE all-reduce of the gradient buffer is the whole exchange."""
    offsets, off = {}, 0
    for name, shape in named_shapes:
        n = 1
        for d in shape:
            n *= int(d)
        offsets[name] = off
        off += (n + align - 1) // align * align
    return offsets, off


def all_reduce_mean(buf, group=None):
    """DDP's gradient exchange on the flat buffer: sum over ranks, then / world (in place).  RCCL over xGMI when
    `buf` lives on a GPU (torch.distributed backend "nccl"), gloo on CPU tensors (tests).  No-op without a group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return buf
    world = dist.get_world_size(group)
    if world == 1:
        return buf
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    buf.mul_(1.0 / world)
    return buf


BN_STATS = 2 * 256 + 1        # per rank: mean[256], M2[256] (sum of squared deviations from that mean), frame count


def gather_bn_stats(stats, group=None):
    """SyncBatchNorm forward exchange (LS-EEND/train_dia_simu.py:167): every rank contributes ONE (mean, M2, n) triple of
    its conv-module BatchNorm input; returns the (R, 513) table all ranks then merge identically (Chan's parallel
    variance -- eend_bn_merge_f32).  One all-gather of 2 KB per BatchNorm layer; R = 1 without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return stats.view(1, -1), 1
    world = dist.get_world_size(group)
    flat = torch.empty(world * stats.numel(), dtype=stats.dtype, device=stats.device)
    dist.all_gather_into_tensor(flat, stats.contiguous().view(-1), group=group)
    return flat.view(world, stats.numel()), world


def all_reduce_bn_sums(sums, group=None):
    """SyncBatchNorm backward exchange: the per-channel sums of d_y and d_y * x_hat become global (in place); the input
    gradient of every rank then uses the global means, the weight / bias gradients stay local (DDP averages them)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums
