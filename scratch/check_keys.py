import sys, os
sys.dont_write_bytecode=True
sys.path.insert(0, "/root/repo")
import torch
exec(open("scratch/mk_fs.py").read())
import fs_eend_amd.fs_model as FM
torch.manual_seed(0)
mine = FM.OnlineTransformerDADiarization(n_speakers=None, in_size=345, n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, has_mask=True, max_seqlen=500, dec_dim_feedforward=2048, mask_delay=0)
sd2 = mine.state_dict()
print(len(sd), len(sd2), list(sd.keys())==list(sd2.keys()))
bad=[k for k in sd if not torch.equal(sd[k], sd2[k])]
print("mismatching tensors:", bad[:5], len(bad))
print(sum(p.numel() for p in mine.parameters()))
