import sys
sys.dont_write_bytecode=True
sys.path.insert(0, "/root/reference/LS-EEND")
import torch
from nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_emb_loss_mask import OnlineConformerRetentionDADiarization
torch.manual_seed(0)
m = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, n_units=256, n_heads=4, enc_n_layers=1, dec_n_layers=1, dropout=0.1, max_seqlen=1000, recurrent_chunk_size=500, feed_forward_expansion_factor=4, dec_dim_feedforward=2048, conv_expansion_factor=2, conv_kernel_size=16, half_step_residual=True, conv_delay=9)
for k,v in m.state_dict().items(): print(k, tuple(v.shape))
