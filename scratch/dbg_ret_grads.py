"""GPU debug: per-retention-module gradient buffers of the HIP LS step vs the fp64 oracle (autograd)."""
import sys
sys.path.insert(0, '.')
import torch
from oracle import fixtures as FX, ls_eend_ref as R, train_ls_ref as TL
from tests.helpers import build_ls_mirror
from fs_eend_amd.train_ls import LsTrainStep
from fs_eend_amd.trainer import prepare_labels
from fs_eend_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else "ls_train_clip"
dev = torch.device("cuda:0")
meta, arr = FX.load_case(name)
m = build_ls_mirror(meta)
feats = FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])
labels = FX.make_labels(meta["lengths"], meta["nspk"], meta["lseed"])

# ---- oracle with taps
taps = []
orig = R.retention_chunk
def tapped(qr, kr, v, L, q=None, role="ret"):
    qr = qr.clone(); kr = kr.clone(); v = v.clone()
    for t in (qr, kr, v): t.retain_grad()
    out = orig(qr, kr, v, L)
    out.retain_grad()
    taps.append((role, qr, kr, v, out))
    return out
R.retention_chunk = tapped
tr = TL.LsTrainRef(m.state_dict(), meta["cfg"], meta["warm"], meta["clip"], False, dtype=torch.float64)
leaves = {k: tr.sd[k].clone().requires_grad_(True) for k in tr.pnames}
sd = dict(tr.sd); sd.update(leaves)
tot, bce, emb, logits, labs = TL.train_loss(sd, [f.double() for f in feats], labels, meta["cfg"], False, {}, torch.float64)
tot.backward()
R.retention_chunk = orig

# ---- HIP with capture
mm = build_ls_mirror(meta).to(dev).train()
eng = LsTrainStep(mm, warmup=meta["warm"], grad_clip=meta["clip"])
eng.prep_weights()
caps = []
orig_bwd = eng._ret_bwd
def cap(bf, g32, ds16, sv, x_in16, nseq, M, wkey, pfx, **kw):
    orig_bwd(bf, g32, ds16, sv, x_in16, nseq, M, wkey, pfx, **kw)
    torch.cuda.synchronize()
    caps.append((pfx, nseq, bf.dqkvg[:M].float().cpu().clone(), bf.ot[:M * 256].float().cpu().clone().view(M, 256), sv.rc.cpu().clone()))
eng._ret_bwd = cap
lens = meta["lengths"]
pl = prepare_labels([l.to(dev) for l in labels], lens)
bf = eng.forward([f.to(dev) for f in feats], pl, lens)
eng.backward(bf)
torch.cuda.synchronize()
B, T, Tp, C = bf.shape
Tv = bf.Tv
print("loss", float(bf.loss[0]), float(bce), float(bf.loss[1]), float(emb))
caps = caps[::-1]          # forward order: encoder layers, decoder layers
assert len(caps) == len(taps)
for (pfx, nseq, dq, ot, rc), (role, qr, kr, v, out) in zip(caps, taps):
    N, H, Tt, d = qr.shape
    def lay(g):            # (N,H,T,d) -> (N, T, H*d)
        return g.transpose(1, 2).reshape(N, Tt, H * d)
    gq, gk, gv = lay(qr.grad), lay(kr.grad), v.grad
    hip = dq.view(nseq, Tp, 4, 256)[:, :Tt].double()
    for nm, e, gidx, sc in (("dq", gq, 0, 1.0), ("dk", gk, 1, 1.0), ("dv", gv, 2, 1.0)):
        h = hip[:, :, gidx]
        if nm == "dk":
            h = h / 0.125                   # HIP column = sk * d/d(k_scaled); the oracle's kr is the scaled k
        err = float((h - e).norm() / e.norm())
        fit = float((h * e).sum() / (e * e).sum())
        cs_h, cs_e = h.sum(dim=(0, 1)), e.sum(dim=(0, 1))
        il = meta["lengths"]
        rep = nseq // len(il)
        mask = torch.zeros(nseq, Tt, dtype=torch.bool)
        for b in range(nseq):
            mask[b, :il[b // rep]] = True
        pad_h = float(h[~mask].norm()) if (~mask).any() else 0.0
        pad_e = float(e[~mask].norm()) if (~mask).any() else 0.0
        print(f"{pfx[-30:]:30s} {nm}: rel L2 {err:.2e} fit {fit-1:+.2e} norm {float(h.norm()/e.norm())-1:+.2e} | colsum norm ratio {float(cs_h.norm()/cs_e.norm())-1:+.2e} "
              f"colsum rel err {float((cs_h-cs_e).norm()/cs_e.norm()):.2e} | pad rows |hip| {pad_h:.2e} |ref| {pad_e:.2e} of {float(e.norm()):.2e}")
    if "dec.layers.1" in pfx:
        for nm, e, gidx in (("dq", gq, 0), ("dk", gk, 1)):
            h = hip[:, :, gidx] / (0.125 if nm == "dk" else 1.0)
            cs_e = e.sum(dim=(0, 1)); chat = cs_e / cs_e.norm()
            contrib = ((h - e) @ chat)            # (nseq, Tt)
            ref_c = (e @ chat)
            print(nm, "total coherent err", float(contrib.sum()), "of", float(ref_c.sum()))
            per_t = contrib.sum(0); ref_t = ref_c.sum(0)
            top = torch.argsort(per_t.abs(), descending=True)[:12]
            print("   top frames:", [(int(t), f"{float(per_t[t]):+.2e}", f"ref {float(ref_t[t]):+.2e}") for t in top])
            per_s = contrib.sum(1)
            print("   per sequence:", [f"{float(x):+.2e}" for x in per_s], " ref ", [f"{float(x):+.2e}" for x in ref_c.sum(1)])
    # o~ check: oracle d out (N,T,H,d) * c_t ; c_t not available directly -> compare direction via dv instead
