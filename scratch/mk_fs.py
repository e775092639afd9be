sys.path.insert(0, "/root/reference/FS-EEND")
from nnet.model.onl_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import OnlineTransformerDADiarization
torch.manual_seed(0)
m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, has_mask=True, max_seqlen=500, dec_dim_feedforward=2048, mask_delay=0).eval()
g = torch.Generator().manual_seed(777)
T=500
src = [torch.randn(T,345,generator=g)*2-3, (torch.randn(T-37,345,generator=g)*2-3)]
ilens=[T, T-37]
sd = m.state_dict()
