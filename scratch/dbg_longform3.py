import sys, torch
sys.path.insert(0, '.')
from oracle import fixtures as FX
from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization as M
dev = torch.device('cuda:0')
base = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, max_seqlen=1000, recurrent_chunk_size=500,
            feed_forward_expansion_factor=4, dec_dim_feedforward=2048, conv_expansion_factor=2, conv_kernel_size=16,
            half_step_residual=True, conv_delay=9)
def run(T, C, **over):
    cfg = dict(base); cfg.update(over)
    torch.manual_seed(0)
    m = M(n_speakers=None, in_size=345, **cfg).eval(); FX.perturb_(m, 5); m = m.to(dev)
    src = [s.to(dev) for s in FX.make_src([T], 345, 4321)]
    a = m.test_chunked(src, [T], C); b = m.test(src, [T], C)
    out = []
    for name, x, y in (("logits", a[0][0], b[0][0]), ("emb", a[1][0], b[1][0]), ("attr", a[2][0].flatten(1), b[2][0].flatten(1))):
        d = (x - y).abs().max(dim=1)[0]
        nz = torch.nonzero(d > 0).flatten()
        out.append(f"{name}: {nz.numel()} frames differ, first {nz[:5].tolist()}, max {d.max().item():.1e}")
    print(f"T={T} C={C} {over}: " + " | ".join(out))
run(36000, 10)
run(16000, 10)
run(16000, 5)
run(16000, 10, dec_dim_feedforward=512)
run(16000, 10, enc_n_layers=1, dec_n_layers=1)
run(16000, 8)
