"""ORACLE-side test infrastructure (never imported by the product path).

Deterministic helpers shared by oracle/gen_golden.py (which runs the *reference* here,
in the build container) and by tests/ (which run the oracle and the HIP path on the
GPU box, where /root/reference does not exist).

Full-size weights (40-45 MB) are too big to commit, so a golden case stores: the seed,
the config, the inputs' seed, the expected outputs, and a checksum of every parameter
tensor.  Both sides rebuild the parameters as ``default init under torch.manual_seed(seed)``
followed by ``perturb_`` (so LayerNorm / BatchNorm affines, biases and running statistics
are non-trivial and the deep-copied layers differ); the checksums prove the rebuilt
parameters are the ones the reference produced the outputs with.
"""
import json
import os
from typing import Dict, List

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


@torch.no_grad()
def perturb_(module: torch.nn.Module, seed: int) -> None:
    """Seeded in-place perturbation of every parameter / BN buffer, in state_dict order."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for name, t in sd.items():
        if not t.is_floating_point():
            continue
        leaf = name.split(".")[-1]
        if name.endswith("pos_enc.pe") or leaf in ("angle", "decay"):
            continue                                            # fixed tables
        r = torch.randn(t.shape, generator=g)
        if leaf == "running_mean":
            t.copy_(0.5 * r - 3.0 if "enc.bn" in name else 0.1 * r)
        elif leaf == "running_var":
            u = torch.rand(t.shape, generator=g)
            t.copy_(4.0 * (0.5 + u) if "enc.bn" in name else 0.5 + u)
        elif t.dim() == 1 and "norm" in name and leaf == "weight":
            t.copy_(1.0 + 0.2 * r)
        elif t.dim() == 1 and leaf == "weight":                 # BatchNorm weight
            t.copy_(1.0 + 0.2 * r)
        elif t.dim() == 1:                                      # biases
            t.add_(0.05 * r)
        else:
            s = t.std() if t.numel() > 1 else torch.tensor(1.0)
            t.add_(0.5 * s * r)


def make_src(lengths: List[int], in_size: int, seed: int) -> List[torch.Tensor]:
    """Synthetic spliced-log-mel-like features: randn * 2 - 3."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(T, in_size, generator=g) * 2 - 3 for T in lengths]


def make_labels(lengths: List[int], n_cols: List[int], seed: int) -> List[torch.Tensor]:
    """Bernoulli(0.3) activity held for 20-frame runs, (T_i, n_cols_i) float32."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for T, c in zip(lengths, n_cols):
        runs = (T + 19) // 20
        a = (torch.rand(runs, c, generator=g) < 0.3).float()
        out.append(a.repeat_interleave(20, dim=0)[:T].contiguous())
    return out


def param_checksums(sd: Dict[str, torch.Tensor]) -> Dict[str, List[float]]:
    """Two position-sensitive float64 checksums per tensor."""
    out = {}
    for k, v in sd.items():
        if not v.is_floating_point():
            continue
        x = v.detach().double().flatten()
        w = torch.arange(1, x.numel() + 1, dtype=torch.float64) % 977 + 1.0
        out[k] = [float(x.abs().sum()), float((x * w).sum())]
    return out


def check_params(sd: Dict[str, torch.Tensor], want: Dict[str, List[float]], rtol=1e-7) -> None:
    got = param_checksums(sd)
    assert set(got) == set(want), f"state_dict keys differ: {set(got) ^ set(want)}"
    for k in want:
        # sin/cos/pow tables are recomputed by libm on the local CPU: equal to ~1 ulp, not bit-equal
        tol = 1e-6 if (k.endswith("pos_enc.pe") or k.split(".")[-1] in ("angle", "decay")) else rtol
        for a, b in zip(got[k], want[k]):
            assert abs(a - b) <= tol * max(1.0, abs(b)), f"parameter {k} is not the golden one ({a} vs {b})"


def save_case(name: str, meta: dict, arrays: Dict[str, np.ndarray]) -> str:
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    return path


def load_case(name: str):
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    z = np.load(path)
    meta = json.loads(bytes(z["__meta__"]).decode())
    arrays = {k: z[k] for k in z.files if k != "__meta__"}
    return meta, arrays


def list_cases(prefix: str) -> List[str]:
    if not os.path.isdir(GOLDEN_DIR):
        return []
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith(prefix) and f.endswith(".npz"))
