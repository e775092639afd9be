import sys, os, time
os.environ["PYTHONDONTWRITEBYTECODE"]="1"
sys.dont_write_bytecode=True
sys.path.insert(0, "/root/reference/FS-EEND")
sys.path.insert(0, "/root/repo")
import torch
from nnet.model.onl_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import OnlineTransformerDADiarization
from oracle import fs_eend_ref as R
torch.manual_seed(0)
m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, has_mask=True, max_seqlen=500, dec_dim_feedforward=2048, mask_delay=0).eval()
g = torch.Generator().manual_seed(777)
T=500
src = [torch.randn(T,345,generator=g)*2-3, (torch.randn(T-37,345,generator=g)*2-3)]
ilens=[T, T-37]
sd = m.state_dict()
with torch.no_grad():
    t0=time.time(); ref = m.test(src, ilens, 6); t1=time.time()
    mine = R.fs_test(src, ilens, sd, n_heads=4, enc_n_layers=4, dec_n_layers=2, max_nspks=6); t2=time.time()
    m64 = R.fs_test(src, ilens, sd, n_heads=4, enc_n_layers=4, dec_n_layers=2, max_nspks=6, dtype=torch.float64)
print("ref time", t1-t0, "oracle", t2-t1)
for name, a, b, c in zip(["logits","emb","attr"], ref, mine, m64):
    for x,y,z in zip(a,b,c):
        print(name, "ref-vs-oracle32", (x-y).abs().max().item(), "ref-vs-f64", (x-z).abs().max().item(), "oracle32-vs-f64", (y-z).abs().max().item())

def bf(x): return x.to(torch.bfloat16).to(x.dtype)
def split(x):
    hi = x.to(torch.bfloat16).to(x.dtype); lo=(x-hi).to(torch.bfloat16).to(x.dtype); return hi+lo
def mk(roles_bf, roles_split=()):
    def q(x, role):
        for r in roles_split:
            if role.startswith(r): return split(x)
        for r in roles_bf:
            if role.startswith(r): return bf(x)
        return x
    return q
cfgs = {
 "all bf16": mk(["enc","dec","cnn","head"]),
 "all bf16 but head": mk(["enc","dec","cnn"]),
 "attn only bf16 (q,k,p,v)": lambda x,role: bf(x) if role.split(".")[-1] in ("q","k","p","v") else x,
 "attn qk only": lambda x,role: bf(x) if role.split(".")[-1] in ("q","k") else x,
 "attn pv only": lambda x,role: bf(x) if role.split(".")[-1] in ("p","v") else x,
 "enc bf16 only": mk(["enc"]),
 "dec bf16 only": mk(["dec"]),
 "cnn bf16 only": mk(["cnn"]),
 "ffn only": lambda x,role: bf(x) if ("ff1" in role or "ff2" in role) else x,
 "all split-bf16": mk([],["enc","dec","cnn","head"]),
 "weights bf16 only": lambda x,role: bf(x) if role.endswith(".w") else x,
 "acts bf16 only": lambda x,role: bf(x) if not role.endswith(".w") else x,
}
with torch.no_grad():
  for k,q in cfgs.items():
    out = R.fs_test(src, ilens, sd, n_heads=4, enc_n_layers=4, dec_n_layers=2, max_nspks=6, q=q)
    e = max((x-y).abs().max().item() for x,y in zip(out[0], m64[0]))
    ee = max((x-y).abs().max().item() for x,y in zip(out[1], m64[1]))
    print(f"{k:32s} logits maxerr {e:.2e}  emb maxerr {ee:.2e}")
