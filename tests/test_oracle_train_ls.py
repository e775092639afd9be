"""CPU: the LS-EEND training-step oracle (oracle/train_ls_ref.py) against the golden vectors the reference itself
produced (oracle/gen_golden_train_ls.py: the reference's training_step, standard_loss / pit_loss_multispk, model in
train() mode, Adam, NoamScheduler, clip_grad_norm_)."""
import numpy as np
import pytest
import torch

from oracle import fixtures as FX
from oracle import train_ls_ref as TL
from tests.helpers import build_ls_mirror

import os
# ls_train_b64 (the bench's batch size; 64 x T=1000 through the fp64-free CPU oracle: 2+ minutes) only with EEND_SLOW_TESTS=1 -- the GPU
# test tests/test_train_step_ls.py runs it always; measured once here: passes (round 5)
CASES = [c for c in FX.list_cases("ls_train_") if c != "ls_train_b64" or os.environ.get("EEND_SLOW_TESTS") == "1"]


def _slice_index(numel, n=24):
    a = np.arange(min(12, numel))
    b = (np.arange(12) * 7919 + 13) % numel
    return np.concatenate([a, b]).astype(np.int64)[:n]


def test_cases_present():
    assert {"ls_train_small", "ls_train_clip", "ls_train_pit", "ls_train_full"} <= set(CASES)


@pytest.mark.parametrize("name", CASES)
def test_ls_train_oracle_vs_reference(name):
    meta, arr = FX.load_case(name)
    if name == "ls_train_full":
        torch.set_num_threads(max(torch.get_num_threads(), 4))
    m = build_ls_mirror(meta)
    assert [n for n, _ in m.named_parameters()] == meta["param_names"]
    feats = FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])
    labels = FX.make_labels(meta["lengths"], meta["nspk"], meta["lseed"])
    tr = TL.LsTrainRef(m.state_dict(), meta["cfg"], meta["warm"], meta["clip"], meta["pit"])
    assert tr.pnames == meta["param_names"]
    for s in range(meta["steps"]):
        out = tr.step(feats, labels)
        want = arr[f"s{s}_loss"]
        assert abs(out["loss"] - want[0]) < 3e-6 and abs(out["bce"] - want[1]) < 3e-6 and abs(out["emb"] - want[2]) < 3e-6
        assert abs(out["lr"] - arr[f"s{s}_lr"][0]) <= 1e-12 + 1e-9 * arr[f"s{s}_lr"][0]
        assert abs(out["gradnorm"] - arr[f"s{s}_gradnorm"][0]) < 3e-4 * arr[f"s{s}_gradnorm"][0]
        if s == 0:
            for i, k in enumerate(meta["param_names"]):
                g = out["grads"][k]
                if k in meta["nograd"]:
                    assert g is None and TL.never_graded(k)
                    continue
                assert not TL.never_graded(k)
                gn = float(g.double().norm())
                assert abs(gn - arr["grad_norms"][i]) < 3e-4 * arr["grad_norms"][i] + 1e-9, k
                idx = _slice_index(g.numel())
                got = g.flatten()[torch.as_tensor(idx)].numpy()
                assert np.abs(got - arr["grad_slices"][i][:len(idx)]).max() < 3e-4 * max(arr["grad_norms"][i], 1e-6), k
        for i, k in enumerate(meta["param_names"]):                       # parameters after the optimiser step
            idx = _slice_index(tr.sd[k].numel())
            got = tr.sd[k].flatten()[torch.as_tensor(idx)].numpy()
            assert np.abs(got - arr[f"s{s}_param_slices"][i][:len(idx)]).max() < 3e-5, (k, s)
        for j, k in enumerate(meta["bn_keys"]):                           # conv-module BatchNorm running statistics
            assert np.abs(tr.sd[k].numpy() - arr[f"s{s}_bn"][j]).max() < 1e-4, (k, s)


def test_retention_scales_carry_no_gradient():
    """retention.py:163,180: inner_scale / kv_scale are .detach()ed -- scaling q by 2 must double d out / d v paths only
    through the un-normalised products, i.e. the oracle's gradient equals that of out = c_t * q_t . sum k (x) v with c_t
    held constant."""
    from oracle import ls_eend_ref as R
    torch.manual_seed(0)
    N, H, T, d, L = 2, 2, 24, 8, 8
    q = torch.randn(N, H, T, d, dtype=torch.float64, requires_grad=True)
    k = torch.randn(N, H, T, d, dtype=torch.float64, requires_grad=True)
    v = torch.randn(N, T, H * d, dtype=torch.float64, requires_grad=True)
    w = torch.randn(N, T, H, d, dtype=torch.float64)
    out = R.retention_chunk(q * 3, k * 3, v, L)           # large scores: the clamp(min=1) branches are active
    gq, gk, gv = torch.autograd.grad((out * w).sum(), [q, k, v])
    # closed form with detached c_t = out / (q_t . prefix state)
    with torch.no_grad():
        qq, kk = q * 3, k * 3
        vv = v.reshape(N, T, H, d).transpose(1, 2)
        causal = torch.tril(torch.ones(T, T, dtype=torch.float64))
        raw = ((qq @ kk.transpose(-1, -2)) * causal) @ vv                          # (N,H,T,d)
        c = (out.transpose(1, 2) / raw)[..., :1]                                   # the detached per-row scalar
        assert torch.allclose(out.transpose(1, 2), c * raw, rtol=1e-9, atol=1e-12)
        wo = w.transpose(1, 2) * c                                                 # o~ = c_t * d out
        A = (wo @ vv.transpose(-1, -2)) * causal
        assert torch.allclose(gq, 3 * (A @ kk), rtol=1e-8, atol=1e-10)
        assert torch.allclose(gk, 3 * (A.transpose(-1, -2) @ qq), rtol=1e-8, atol=1e-10)
        S = (qq @ kk.transpose(-1, -2)) * causal
        assert torch.allclose(gv.reshape(N, T, H, d).transpose(1, 2), S.transpose(-1, -2) @ wo, rtol=1e-8, atol=1e-10)
