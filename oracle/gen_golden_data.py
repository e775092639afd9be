"""Golden vectors for the data-side helpers -- runs ONLY in the build container.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_data.py

`MyDistributedSampler` (LS-EEND/data_loaders/utils/my_distributed_sampler.py; the module imports pytorch_lightning's
rank_zero_warn, absent here, so the class definition is evaluated out of the file with that one name bound to
warnings.warn) and `_count_frames` / `_gen_frame_indices` (datasets/diarization_dataset_on_the_fly.py) are run on a
grid of cases; the on-the-fly chunk start (`rng.choice(range(data_len))` on PCG64(seed), :94-98) is evaluated with the
same numpy calls.  Only inputs and outputs are stored (tests/golden/data_*.json).
"""
import ast
import json
import math
import os
import sys
import warnings
from typing import Iterator, Optional

sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.distributed import DistributedSampler
from typing import TypeVar
T_co = TypeVar("T_co", covariant=True)

REF = "/root/reference/LS-EEND"
OUT = os.path.join(ROOT, "tests", "golden")


def defs(path, names):
    tree = ast.parse(open(path).read())
    ns = {"math": math, "Iterator": Iterator, "Optional": Optional, "torch": torch, "Dataset": Dataset,
          "DistributedSampler": DistributedSampler, "T_co": T_co, "rank_zero_warn": warnings.warn, "np": np}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def main():
    (Sampler,) = defs(f"{REF}/data_loaders/utils/my_distributed_sampler.py", ["MyDistributedSampler"])
    cf, gfi = defs(f"{REF}/datasets/diarization_dataset_on_the_fly.py", ["_count_frames", "_gen_frame_indices"])
    samp = []
    for n, world, shuffle, seed, drop_last in [(10, 1, True, 0, False), (10, 4, True, 3, False), (10, 4, True, 3, True),
                                                (7, 2, False, 5, False), (3, 8, True, 1, False), (101, 8, True, 777, False)]:
        for epoch in (0, 1, 5):
            per_rank = []
            for rank in range(world):
                s = Sampler(list(range(n)), num_replicas=world, rank=rank, shuffle=shuffle, seed=seed, drop_last=drop_last)
                s.set_epoch(epoch)
                per_rank.append([[int(i), int(sd)] for i, sd in s])
                assert len(per_rank[-1]) == len(s)
            samp.append(dict(n=n, world=world, shuffle=shuffle, seed=seed, drop_last=drop_last, epoch=epoch, pairs=per_rank))
    grid = []
    for data_len in (0, 1, 499, 500, 501, 999, 1000, 1234, 2000, 5003):
        for size, step in ((500, 500), (500, 250), (2000, 1000), (1000, 1000)):
            for uls in (False, True):
                for delay in (0, 5):
                    grid.append(dict(data_len=data_len, size=size, step=step, use_last_samples=uls, label_delay=delay,
                                     count=cf(data_len, size, step),
                                     chunks=[[int(a), int(b)] for a, b in gfi(data_len, size, step, uls, label_delay=delay)]))
    otf = []
    for seed in (0, 1, 123456789, 9999999998):
        for data_len in (1, 37, 10000, 123457):
            rng = np.random.default_rng(np.random.PCG64(seed))
            st = int(rng.choice(range(data_len)))
            otf.append(dict(seed=seed, data_len=data_len, chunk_size=1000, subsampling=10, st=st, ed=min(st + 1000 * 10, data_len)))
    with open(os.path.join(OUT, "data_sampler.json"), "w") as f:
        json.dump(dict(sampler=samp, grid=grid, on_the_fly=otf, torch=torch.__version__, numpy=np.__version__), f)
    print(len(samp), "sampler cases,", len(grid), "grid cases,", len(otf), "on-the-fly cases")


if __name__ == "__main__":
    main()
