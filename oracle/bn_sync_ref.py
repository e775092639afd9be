"""ORACLE (test infrastructure, NOT product code) -- the arithmetic of the SyncBatchNorm exchange of the LS-EEND
training step (LS-EEND/train_dia_simu.py:167 `sync_batchnorm`; torch.nn.SyncBatchNorm semantics): per-rank
(mean, M2, n) triples merged with Chan's parallel-variance formula, and the backward with global per-channel sums.
Restates what csrc/ls_train.hip bn_colstats16 / bn_merge / bn_swish_bwd_* compute, in float64 torch on the CPU.
Only tests/ import this file."""
import torch


def local_stats(c: torch.Tensor) -> torch.Tensor:
    """c: (rows, 256) valid frames of this rank -> [mean(256), M2(256), n]."""
    c = c.double()
    mean = c.mean(0)
    return torch.cat([mean, ((c - mean) ** 2).sum(0), torch.tensor([float(c.shape[0])], dtype=torch.float64)])


def merge(table: torch.Tensor):
    """(R, 513) -> mean, biased var, n, unbiased var (running-statistics update)."""
    D = (table.shape[1] - 1) // 2
    n_r = table[:, -1:].double()
    n = n_r.sum()
    mean = (n_r * table[:, :D].double()).sum(0) / n
    m2 = (table[:, D:2 * D].double() + n_r * (table[:, :D].double() - mean) ** 2).sum(0)
    return mean, m2 / n, float(n), m2 / (n - 1)


def swish_grad(y):
    s = torch.sigmoid(y)
    return s * (1 + y * (1 - s))


def bwd_sums(ds, c, mean, var, gamma, beta, eps=1e-5):
    """Per-channel S1 = sum d_y, S2 = sum d_y * c_hat of this rank's rows (d_y = ds * swish'(BN(c)))."""
    ch = (c.double() - mean) / torch.sqrt(var + eps)
    dy = ds.double() * swish_grad(gamma * ch + beta)
    return torch.cat([dy.sum(0), (dy * ch).sum(0)])


def bwd_apply(ds, c, mean, var, gamma, beta, sums, n, eps=1e-5):
    """d_c of this rank's rows given the GLOBAL sums and frame count."""
    D = c.shape[1]
    rs = 1.0 / torch.sqrt(var + eps)
    ch = (c.double() - mean) * rs
    dy = ds.double() * swish_grad(gamma * ch + beta)
    return gamma * rs * (dy - sums[:D] / n - ch * sums[D:] / n)
