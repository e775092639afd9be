import sys, torch
sys.path.insert(0, '.')
from fs_eend_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
D, K, Tp = 256, 16, 128
x = torch.randn(1, 2 * Tp, D, device=dev, generator=g).to(torch.float16)
w = torch.randn(D, K, device=dev, generator=g) * 0.2
bn = tuple(t.contiguous() for t in (torch.ones(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.ones(D, device=dev)))
full = torch.empty(2 * Tp, D, dtype=torch.float16, device=dev)
ops.dwconv_bn_swish(x.view(-1, D), w, bn, full, 1, 2 * Tp)
second = torch.empty(Tp, D, dtype=torch.float16, device=dev)
halo = x[:, Tp - (K - 1):Tp].contiguous()
ops.dwconv_bn_swish(x[:, Tp:].contiguous().view(-1, D), w, bn, second, 1, Tp, halo16=halo)
d = (second.float() - full[Tp:].float()).abs().max(dim=1)[0]
print("with halo: max diff per frame (first 20):", [f"{v:.1e}" for v in d[:20].tolist()])
nohalo = torch.empty(Tp, D, dtype=torch.float16, device=dev)
ops.dwconv_bn_swish(x[:, Tp:].contiguous().view(-1, D), w, bn, nohalo, 1, Tp)
d = (nohalo.float() - full[Tp:].float()).abs().max(dim=1)[0]
print("no halo  : max diff per frame (first 20):", [f"{v:.1e}" for v in d[:20].tolist()])
