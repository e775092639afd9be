import sys, torch
sys.path.insert(0, '.')
from oracle import fixtures as FX
from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization as M
from fs_eend_amd import ops
dev = torch.device('cuda:0')
cfg = dict(n_units=256, n_heads=4, enc_n_layers=1, dec_n_layers=1, dropout=0.1, max_seqlen=1000, recurrent_chunk_size=500,
           feed_forward_expansion_factor=4, dec_dim_feedforward=512, conv_expansion_factor=2, conv_kernel_size=16,
           half_step_residual=True, conv_delay=9)
torch.manual_seed(0)
m = M(n_speakers=None, in_size=345, **cfg).eval(); FX.perturb_(m, 5); m = m.to(dev)
T = 16000
src = [s.to(dev) for s in FX.make_src([T], 345, 1)]
a = m.test_chunked(src, [T], 4)
b = m.test(src, [T], 4)
for name, x, y in (("logits", a[0][0], b[0][0]), ("emb", a[1][0], b[1][0])):
    d = (x - y).abs()
    print(name, "max diff per 1000 frames:", [f"{d[i:i+1000].max().item():.1e}" for i in range(0, T, 1000)])
# determinism of the monolithic path itself
c = m.test(src, [T], 4)
print("monolithic run-to-run equal:", torch.equal(b[0][0], c[0][0]), torch.equal(b[1][0], c[1][0]))
a2 = m.test_chunked(src, [T], 4)
print("chunked run-to-run equal:", torch.equal(a[0][0], a2[0][0]))
# monolithic on the first 8000 frames only vs first 8000 of the 16000 run (causality => equal if tiling-invariant)
b8 = m.test([src[0][:8000]], [8000], 4)
d = (b8[1][0][:7990] - b[1][0][:7990]).abs().max().item()
print("monolithic T=8000 vs T=16000 prefix (emb, first 7990):", d)
d = (b8[0][0][:7990] - b[0][0][:7990]).abs().max().item()
print("monolithic T=8000 vs T=16000 prefix (logits, first 7990):", d)
