"""CPU: the PIT label-assignment oracle (oracle/pit_ref.py) against golden outputs of the reference's own
batch_pit_n_speaker_loss / pit_loss_multispk bodies (tests/golden/pit_*.npz, oracle/gen_golden_pit.py)."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import gen_golden_pit as G
from oracle import pit_ref as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return ast.literal_eval(str(z["meta"])), z


@pytest.mark.parametrize("case", [c["name"] for c in G.CASES])
def test_pit_oracle_vs_reference_golden(case):
    meta, z = load(case)
    ys, ts = G.pit_inputs(meta)
    loss, labels = P.batch_pit_n_speaker_loss(ys, ts, list(meta["nspk"]))
    assert abs(float(loss) - float(z["loss"][0])) <= 1e-5 * max(1.0, abs(float(z["loss"][0])))
    for i, l in enumerate(labels):
        assert np.array_equal(l.numpy().astype(np.int8), z[f"bpit_label{i}"])
    perm = P.pit_loss_multispk(ys, ts, list(meta["nspk"]))
    for i, l in enumerate(perm):
        assert np.array_equal(l.numpy().astype(np.int8), z[f"multi_label{i}"])


def test_cost_matrix_is_the_bce_sum():
    g = torch.Generator().manual_seed(0)
    y, t = torch.randn(40, 3, generator=g), (torch.rand(40, 3, generator=g) < 0.5).float()
    c = P.cost_matrices([y], [t])[0]
    for i in range(3):
        for j in range(3):
            want = torch.nn.functional.binary_cross_entropy_with_logits(y[:, i], t[:, j], reduction="sum")
            assert abs(float(c[i, j]) - float(want)) < 1e-4
