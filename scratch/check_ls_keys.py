import sys
sys.dont_write_bytecode=True
sys.path.insert(0, "/root/repo")
import torch
import fs_eend_amd.ls_model as LM
cfg = dict(n_units=256, n_heads=4, enc_n_layers=2, dec_n_layers=2, dropout=0.1, max_seqlen=1000, recurrent_chunk_size=500, feed_forward_expansion_factor=4, dec_dim_feedforward=2048, conv_expansion_factor=2, conv_kernel_size=16, half_step_residual=True, conv_delay=9)
torch.manual_seed(0)
mine = LM.OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **cfg)
sd2 = {k: v.clone() for k,v in mine.state_dict().items()}
sys.path.insert(0, "/root/reference/LS-EEND")
from nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_emb_loss_mask import OnlineConformerRetentionDADiarization
torch.manual_seed(0)
m = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **cfg)
sd = m.state_dict()
print(len(sd), len(sd2), list(sd.keys())==list(sd2.keys()))
print([k for k in sd if k not in sd2][:5], [k for k in sd2 if k not in sd][:5])
bad=[k for k in sd if k in sd2 and not torch.equal(sd[k], sd2[k])]
print("mismatching tensors:", bad[:5], len(bad))
