import sys, os
sys.dont_write_bytecode=True
sys.path.insert(0, "/root/repo")
import torch
from oracle import ls_eend_ref as R, fixtures as FX
sys.path.insert(0, "/root/reference/LS-EEND")
from nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_emb_loss_mask import OnlineConformerRetentionDADiarization
cfg = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, max_seqlen=1000, recurrent_chunk_size=500, feed_forward_expansion_factor=4, dec_dim_feedforward=2048, conv_expansion_factor=2, conv_kernel_size=16, half_step_residual=True, conv_delay=9)
torch.manual_seed(20)
m = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **cfg).eval()
FX.perturb_(m, 31)
sd = m.state_dict()
src = FX.make_src([1500], 345, 801); ilens=[1500]
kw = dict(n_heads=4, enc_n_layers=4, dec_n_layers=2, max_nspks=6)
def bf(x): return x.to(torch.bfloat16).to(x.dtype)
def hf(x): return x.to(torch.float16).to(x.dtype)
def split(x):
    hi = bf(x); return hi + bf(x-hi)
ret = ("q","k","p","v","s","qc")
def mk(pmode, smode):
    def q(x, role):
        leaf = role.split(".")[-1]
        if leaf in ret and ".ret" in role:
            if leaf == "p": return pmode(x)
            if leaf == "s": return smode(x)
            return bf(x)
        if leaf in ("q","k","p","v"): return bf(x)   # speaker MHA (VALU fp32 in product, but inputs f16) 
        return hf(x)
    return q
with torch.no_grad():
    m64 = R.ls_test(src, ilens, sd, dtype=torch.float64, **kw)
    ref = m.test(src, ilens, 6)
    print("ref fp32 vs f64 logits", (ref[0][0]-m64[0][0]).abs().max().item())
    for name,(pm,sm) in {"p bf16, state bf16": (bf,bf), "p bf16, state split": (bf,split), "p split, state split": (split,split), "p f16, state split": (hf, split)}.items():
        out = R.ls_test(src, ilens, sd, q=mk(pm,sm), **kw)
        print(f"{name:28s} logits maxerr {(out[0][0]-m64[0][0]).abs().max().item():.2e}")
    out = R.ls_test(src, ilens, sd, q=lambda x,r: hf(x) if r.split('.')[-1] not in ret+("q","k","p","v") else x, **kw)
    print(f"{'linears f16 only':28s} logits maxerr {(out[0][0]-m64[0][0]).abs().max().item():.2e}")
def splith(x):
    hi = hf(x); return hi + hf(x-hi)
def mk2(qkv, pmode, smode, lin=hf):
    def q(x, role):
        leaf = role.split(".")[-1]
        if leaf in ret and ".ret" in role:
            if leaf == "p": return pmode(x)
            if leaf == "s": return smode(x)
            return qkv(x)
        if leaf in ("q","k","p","v"): return x
        return lin(x)
    return q
with torch.no_grad():
    for name,args in {"ret all f16, state split-f16": (hf,hf,splith), "ret f16, p split, state split": (hf,splith,splith), "ret qkv split-f16,p f16": (splith,hf,splith), "ret exact, lin f16": (lambda x:x,)*3,
                      "ret f16/state split, lin exact": (hf,hf,splith,lambda x:x)}.items():
        out = R.ls_test(src, ilens, sd, q=mk2(*args), **kw)
        print(f"{name:34s} logits maxerr {(out[0][0]-m64[0][0]).abs().max().item():.2e}")
