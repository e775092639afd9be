import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
dev = torch.device("cuda"); F16, F32 = torch.float16, torch.float32
def rnd(shape, seed, dtype=F32, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)
M, Fh = 128, 2048
a = rnd((M, 256), 21, F16); wo, bo = rnd((256, 256), 22, F16, 0.06), rnd((256,), 23) * 0.2
w1, b1 = rnd((Fh, 256), 24, F16, 0.08), rnd((Fh,), 25) * 0.3
w2, b2 = rnd((256, Fh), 26, F16, 0.04), rnd((256,), 27) * 0.3
res = rnd((M, 256), 28)
one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
def run(w2_, b2_, tag):
    o32 = torch.empty((M, 256), dtype=F32, device=dev); o16 = torch.empty((M, 256), dtype=F16, device=dev)
    ops.attnout_ffn_fused(a, wo, bo, res, one, zero, 1e-5, w1, b1, w2_, b2_, one, zero, 1e-5, o32, o16)
    x = torch.nn.functional.layer_norm(a.float() @ wo.float().t() + bo + res, (256,), one, zero, 1e-5)
    h = (x.to(F16).float() @ w1.float().t() + b1).relu().to(F16).float()
    want = torch.nn.functional.layer_norm(h @ w2_.float().t() + b2_ + x, (256,), one, zero, 1e-5)
    err = (o32 - want).abs()
    print(tag, "max err", err.max().item())
    print(" per 16-row block:", [round(err[r:r+16].max().item(), 4) for r in range(0, M, 16)])
    print(" per 32-col block:", [round(err[:, c:c+32].max().item(), 4) for c in range(0, 256, 32)])
run(torch.zeros_like(w2), torch.zeros_like(b2), "x only (W2=0)")
run(w2, b2, "full")
