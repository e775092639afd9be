"""GPU: device PIT label assignment against the golden outputs of the reference's own functions
(tests/golden/pit_*.npz) and against the oracle on larger random batches.  Permuted labels exact; loss 1e-5 rel."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import gen_golden_pit as G
from oracle import pit_ref as P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return ast.literal_eval(str(z["meta"])), z


@pytest.mark.parametrize("case", [c["name"] for c in G.CASES])
def test_pit_vs_reference_golden(hip_lib, dev, case):
    from fs_eend_amd import pit
    meta, z = load(case)
    ys, ts = G.pit_inputs(meta)
    yd, td = [y.to(dev) for y in ys], [t.to(dev) for t in ts]
    loss, labels = pit.batch_pit_n_speaker_loss(yd, td, list(meta["nspk"]))
    assert abs(float(loss) - float(z["loss"][0])) <= 1e-5 * max(1.0, abs(float(z["loss"][0])))
    for i, l in enumerate(labels):
        assert np.array_equal(l.cpu().numpy().astype(np.int8), z[f"bpit_label{i}"])
    tgt = torch.nn.utils.rnn.pad_sequence(td, padding_value=-1, batch_first=True)
    for i, l in enumerate(pit.pit_loss_multispk(yd, tgt, np.array(meta["nspk"]))):
        assert np.array_equal(l.cpu().numpy().astype(np.int8), z[f"multi_label{i}"])


@pytest.mark.parametrize("B,C,Tmax,seed", [(64, 4, 500, 1), (16, 6, 300, 2), (5, 10, 120, 3), (3, 16, 90, 4), (1, 1, 40, 5)])
def test_pit_vs_oracle_random(hip_lib, dev, B, C, Tmax, seed):
    from fs_eend_amd import pit
    g = torch.Generator().manual_seed(seed)
    lens = [int(torch.randint(Tmax // 2, Tmax + 1, (1,), generator=g)) for _ in range(B)]
    nspk = [int(torch.randint(1, C + 1, (1,), generator=g)) for _ in range(B)]
    nspk[0] = C
    c = dict(seed=seed + 100, lens=lens, nspk=nspk, C=C)
    ys, ts = G.pit_inputs(c)
    yd, td = [y.to(dev) for y in ys], [t.to(dev) for t in ts]
    perm = pit.pit_loss_multispk(yd, td, nspk)
    want = P.pit_loss_multispk(ys, ts, nspk)
    for a, b in zip(perm, want):
        assert torch.equal(a.cpu(), b)
    if C <= 6:                                     # the brute-force oracle enumerates C! permutations
        loss, labels = pit.batch_pit_n_speaker_loss(yd, td, nspk)
        wl, wlab = P.batch_pit_n_speaker_loss(ys, ts, nspk)
        assert abs(float(loss) - float(wl)) <= 1e-5 * max(1.0, abs(float(wl)))
        for a, b in zip(labels, wlab):
            assert torch.equal(a.cpu(), b)
