"""Does bf16 rounding of q/k/v/o~/A/S in the retention backward bias the q/k gradient norms? (emulation on the oracle)"""
import sys, math
sys.path.insert(0, '.')
import torch
from oracle import fixtures as FX, ls_eend_ref as R, train_ls_ref as TL
from tests.helpers import build_ls_mirror

def bf(x): return x.to(torch.bfloat16).to(x.dtype)
def f16(x): return x.to(torch.float16).to(x.dtype)

MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16"

class RetCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, L):
        # q,k: (N,H,T,d); v: (N,T,H*d) -> (N,T,H,d)
        with torch.no_grad():
            out = R._orig_chunk(q, k, v, L)
        ctx.save_for_backward(q, k, v, out)
        ctx.L = L
        return out
    @staticmethod
    def backward(ctx, go):
        q, k, v, out = ctx.saved_tensors
        L = ctx.L
        N, H, T, d = q.shape
        vv = v.reshape(N, T, H, d).transpose(1, 2)
        o = out.transpose(1, 2)
        causal = torch.tril(torch.ones(T, T, dtype=q.dtype))
        raw = ((q @ k.transpose(-1, -2)) * causal) @ vv
        c = (o / raw).nan_to_num(0.0)
        c = c.mean(-1, keepdim=True)  # per-row scalar (all components equal up to rounding)
        ot = go.transpose(1, 2) * c
        if MODE == "bf16":
            qq, kk, v2, ot2 = bf(q), bf(k), bf(vv), bf(ot)
            A = bf((ot2 @ v2.transpose(-1, -2)) * causal)
            S = bf((qq @ kk.transpose(-1, -2)) * causal)
        elif MODE == "f16":
            beta = 16.0 / ot.abs().amax(dim=(-1, -2), keepdim=True)
            qq, kk, v2, ot2 = f16(q), f16(k), f16(vv), f16(ot * beta) / beta
            A = f16((ot2 * beta @ v2.transpose(-1, -2)) * causal) / beta
            S = f16((qq @ kk.transpose(-1, -2)) * causal)
        elif MODE == "bf16_demean":
            qq, kk, ot2 = bf(q), bf(k), bf(ot)
            v2 = bf(vv - vv.mean(-1, keepdim=True))
            A = bf((ot2 @ v2.transpose(-1, -2)) * causal)
            S = bf((qq @ kk.transpose(-1, -2)) * causal)
            v2 = bf(vv)
        else:
            qq, kk, v2, ot2 = q, k, vv, ot
            A = (ot2 @ v2.transpose(-1, -2)) * causal
            S = (qq @ kk.transpose(-1, -2)) * causal
        dq = A @ kk
        dk = A.transpose(-1, -2) @ qq
        dv = S.transpose(-1, -2) @ ot2
        return dq, dk, dv.transpose(1, 2).reshape(N, T, H * d), None

R._orig_chunk = R.retention_chunk
def patched(qr, kr, v, L, q=None, role="ret"):
    return RetCore.apply(qr, kr, v, L)
R.retention_chunk = patched

meta, arr = FX.load_case("ls_train_clip")
m = build_ls_mirror(meta)
feats = FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])
labels = FX.make_labels(meta["lengths"], meta["nspk"], meta["lseed"])
tr = TL.LsTrainRef(m.state_dict(), meta["cfg"], meta["warm"], meta["clip"], meta["pit"], dtype=torch.float64)
tot, bce, emb, grads, bn, _, _ = tr.grads([f.double() for f in feats], labels)
names = meta["param_names"]
import pickle, os
if MODE == "exact":
    pickle.dump({k: (None if g is None else g.clone()) for k, g in grads.items()}, open("/tmp/exact_grads.pkl", "wb"))
ex = pickle.load(open("/tmp/exact_grads.pkl", "rb"))
for i, k in enumerate(names):
    if grads[k] is not None and ("q_proj" in k or "k_proj" in k or "v_proj.weight" in k):
        print(f"  L2err {float((grads[k]-ex[k]).norm()/ex[k].norm()):.2e}  {k}")
for i, k in enumerate(names[:0]):
    if "self_attn1" in k and ("q_proj" in k or "k_proj" in k or "v_proj" in k) or "self_attn.q_proj" in k:
        g = grads[k]
        print(f"{k:60s} {float(g.norm()):.4e} ref {arr['grad_norms'][i]:.4e}  rel {float(g.norm())/arr['grad_norms'][i]-1:+.2e}")
