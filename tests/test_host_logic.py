"""CPU: host-side logic that needs no GPU -- the workspace LRU, optimiser schedule helpers."""
import types

import torch


def test_workspace_cache_lru_and_budget():
    from fs_eend_amd.fs_model import WorkspaceCache
    mk = lambda n: types.SimpleNamespace(a=torch.zeros(n, dtype=torch.uint8), note="x")
    c = WorkspaceCache(budget_bytes=1000, max_entries=3)
    c.put("a", mk(400)); c.put("b", mk(400))
    assert c.get("a") is not None                      # touch: "a" becomes most recent
    c.put("c", mk(400))                                # 1200 > 1000: the least recently used ("b") goes
    assert c.get("b") is None and c.get("a") is not None and c.get("c") is not None
    c.put("big", mk(5000))                             # over budget on its own: everything else is evicted, it stays
    assert len(c) == 1 and c.get("big") is not None
    c.clear()
    c = WorkspaceCache(budget_bytes=10 ** 9, max_entries=2)
    for k in "abc":
        c.put(k, mk(10))
    assert len(c) == 2 and c.get("a") is None


def test_constant_lr_without_scheduler_and_noam_with():
    from fs_eend_amd.train import TrainStepBase, noam_lr
    e = TrainStepBase.__new__(TrainStepBase)
    e.warmup, e.base_lr, e.sched_scale = None, 3e-4, 1.0
    assert [e.current_lr(t) for t in (1, 2, 1000)] == [3e-4] * 3
    e.warmup, e.base_lr = 25, 1.0
    assert e.current_lr(1) == noam_lr(1, 256, 25) and e.current_lr(30) == noam_lr(30, 256, 25)
