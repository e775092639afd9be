exec(open("scratch/prec_ls.py").read().split("def bf(x)")[0])
def hf(x): return x.to(torch.float16).to(x.dtype)
ret = ("q","k","p","v","s","qc")
groups = {"enc.in":["enc.in"], "enc.ffa/ffb":["enc.ffa","enc.ffb"], "enc.ret proj (qp,kp,vp,gp,op)":["enc.ret.qp","enc.ret.kp","enc.ret.vp","enc.ret.gp","enc.ret.op"],
          "enc.ret q/k proj only":["enc.ret.qp","enc.ret.kp"], "enc.ret v/g/out":["enc.ret.vp","enc.ret.gp","enc.ret.op"],
          "conv.pw":["conv.pw"], "cnn":["cnn"], "dec.convert":["dec.convert"], "dec.ret proj":["dec.ret.qp","dec.ret.kp","dec.ret.vp","dec.ret.gp","dec.ret.op"], "dec.mha_s lin":["dec.mha_s.in","dec.mha_s.out"], "dec.ff":["dec.ff"], "head":["head"]}
with torch.no_grad():
    m64 = R.ls_test(src, ilens, sd, dtype=torch.float64, **kw)
    for name, pre in groups.items():
        def q(x, role, pre=pre):
            leaf = role.split(".")[-1]
            if leaf in ret+("q","k","p","v") and not role.endswith((".qp",".kp",".vp",".gp",".op")): return x
            return hf(x) if any(role.startswith(p) for p in pre) else x
        out = R.ls_test(src, ilens, sd, q=q, **kw)
        print(f"{name:34s} logits maxerr {(out[0][0]-m64[0][0]).abs().max().item():.2e}")
