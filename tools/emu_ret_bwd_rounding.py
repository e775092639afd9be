"""CPU study (oracle only): which operand rounding of the retention backward puts the first Conformer block's q / k projection
gradients at ~1e-2 whole-tensor relative error (tests/test_train_step_ls.py)?  The retention is multilinear in (q, k, v) once its
scales are detached, so the HIP backward equals autograd through the same function evaluated on ROUNDED operands with a ROUNDED
incoming gradient.  Modes: bf16 everything (what the kernels do), f16 q/k/v + bf16 gradient, f16 everything (gradient pre-scaled)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fixtures as FX          # noqa: E402
from oracle import ls_eend_ref as R        # noqa: E402
from oracle import train_ls_ref as TL      # noqa: E402
from tests.helpers import build_ls_mirror  # noqa: E402

_exact = R.retention_chunk


def rnd(x, kind, scale=1.0):
    if kind is None:
        return x
    dt = torch.bfloat16 if kind == "bf16" else torch.float16
    return ((x * scale).to(torch.float32).to(dt).to(x.dtype)) / scale


class RetEmu(torch.autograd.Function):
    mode = ("bf16", "bf16", 1.0, None)  # (operand kind, gradient kind, gradient pre-scale, kind the dq / dk / dv results are stored in)

    @staticmethod
    def forward(ctx, qr, kr, v, L):
        ctx.save_for_backward(qr, kr, v)
        ctx.L = L
        with torch.no_grad():
            return _exact(qr, kr, v, L)

    @staticmethod
    def backward(ctx, dout):
        qr, kr, v = ctx.saved_tensors
        ok, gk, gs, outk = RetEmu.mode
        with torch.enable_grad():
            a, b, c = (rnd(t.detach(), ok).requires_grad_(True) for t in (qr, kr, v))
            out = _exact(a, b, c, ctx.L)
            ga, gb, gc = torch.autograd.grad(out, [a, b, c], rnd(dout, gk, gs))
        return rnd(ga, outk, gs), rnd(gb, outk, gs), rnd(gc, outk, gs), None


def patched(qr, kr, v, L, q=None, role="ret"):
    return RetEmu.apply(qr, kr, v, L)


def main():
    meta, _ = FX.load_case("ls_train_small")
    m = build_ls_mirror(meta).train()
    sd = {k: v.detach().double() if v.is_floating_point() else v for k, v in m.state_dict().items()}
    feats = [f.double() for f in FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])]
    raw = [l.double() for l in FX.make_labels(meta["lengths"], meta["nspk"], meta["lseed"])]
    cfg = dict(meta["cfg"], n_units=256)
    pn = [k for k, v in sd.items() if v.is_floating_point() and v.dim() >= 1
          and not k.endswith(("running_mean", "running_var", "pos_enc.pe", ".angle", ".decay"))]

    def grads():
        leaves = {k: sd[k].clone().requires_grad_(True) for k in pn}
        sdd = dict(sd); sdd.update(leaves)
        tot, *_ = TL.train_loss(sdd, feats, raw, cfg, dtype=torch.float64)
        return dict(zip(pn, torch.autograd.grad(tot, [leaves[k] for k in pn], allow_unused=True)))

    exact = grads()
    totn = math.sqrt(sum(float((g ** 2).sum()) for g in exact.values() if g is not None))
    R.retention_chunk = patched
    for mode in (("bf16", "bf16", 1.0, None), ("bf16", "bf16", 1.0, "bf16"), (None, None, 1.0, "bf16"), ("f16", "f16", 16384.0, "bf16"), ("f16", "f16", 16384.0, "f16")):
        RetEmu.mode = mode
        g = grads()
        rows = []
        for k in pn:
            if exact[k] is None or not any(s in k for s in ("q_proj", "k_proj", "v_proj", "g_proj")):
                continue
            err = float((g[k] - exact[k]).norm()) / max(float(exact[k].norm()), 1e-3 * totn)
            rows.append((err, k))
        rows.sort(reverse=True)
        print(f"operands {mode[0]}, gradient {mode[1]} (x{mode[2]:g}), results {mode[3]}:", ", ".join(f"{e:.1e} {k.split('layers.')[-1][:34]}" for e, k in rows[:6]), flush=True)
    R.retention_chunk = _exact


if __name__ == "__main__":
    main()
