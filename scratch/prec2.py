import sys, os
sys.dont_write_bytecode=True
sys.path.insert(0, "/root/repo")
import torch
from oracle import fs_eend_ref as R
exec(open("scratch/mk_fs.py").read())
def bf(x): return x.to(torch.bfloat16).to(x.dtype)
def hf(x): return x.to(torch.float16).to(x.dtype)
attn_roles=("q","k","p","v")
cfgs = {
 "lin fp16, attn bf16": lambda x,role: bf(x) if role.split(".")[-1] in attn_roles else hf(x),
 "all fp16": lambda x,role: hf(x),
 "lin fp16, attn bf16, head fp32": lambda x,role: x if role.startswith("head") else (bf(x) if role.split(".")[-1] in attn_roles else hf(x)),
 "lin fp16 + attn qk bf16, pv fp16": lambda x,role: bf(x) if role.split(".")[-1] in ("q","k") else hf(x),
}
for scale in (1.0, 2.0, 4.0):
  sd2 = {k:(v*scale if (("in_proj_weight" in k) or ("linear" in k and "weight" in k)) else v) for k,v in sd.items()}
  with torch.no_grad():
    m64 = R.fs_test(src, ilens, sd2, n_heads=4, enc_n_layers=4, dec_n_layers=2, max_nspks=6, dtype=torch.float64)
    for k,q in cfgs.items():
      out = R.fs_test(src, ilens, sd2, n_heads=4, enc_n_layers=4, dec_n_layers=2, max_nspks=6, q=q)
      e = max((x-y).abs().max().item() for x,y in zip(out[0], m64[0]))
      print(f"scale {scale} {k:36s} logits maxerr {e:.2e}")
