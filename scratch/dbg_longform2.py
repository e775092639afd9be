import sys, torch
sys.path.insert(0, '.')
from oracle import fixtures as FX
from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization as M
from fs_eend_amd import ops
import fs_eend_amd.ls_model as LM
dev = torch.device('cuda:0')
cfg = dict(n_units=256, n_heads=4, enc_n_layers=1, dec_n_layers=1, dropout=0.1, max_seqlen=1000, recurrent_chunk_size=500,
           feed_forward_expansion_factor=4, dec_dim_feedforward=512, conv_expansion_factor=2, conv_kernel_size=16,
           half_step_residual=True, conv_delay=9)
torch.manual_seed(0)
m = M(n_speakers=None, in_size=345, **cfg).eval(); FX.perturb_(m, 5); m = m.to(dev)
T = 16000
src = [s.to(dev) for s in FX.make_src([T], 345, 1)]
# capture the encoder output of both paths by wrapping conv1d_l2norm
cap = {}
orig = ops.conv1d_l2norm
def spy(x16, *a, **k):
    cap.setdefault("enc", []).append(x16.clone())
    return orig(x16, *a, **k)
ops.conv1d_l2norm = spy
a = m.test_chunked(src, [T], 4)
b = m.test(src, [T], 4)
ea, eb = cap["enc"][0].float(), cap["enc"][1].float()
n = min(ea.shape[0], eb.shape[0])
d = (ea[:n] - eb[:n]).abs().max(dim=1)[0]
nz = torch.nonzero(d[:T] > 0).flatten()
print("encoder-output frames that differ:", nz[:40].tolist(), "count", nz.numel(), "max", d[:T].max().item())
de = (a[1][0] - b[1][0]).abs().max(dim=1)[0]
nz = torch.nonzero(de > 0).flatten()
print("emb frames that differ:", nz[:60].tolist(), "count", nz.numel())
