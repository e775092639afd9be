"""Test-side helpers: rebuild golden-case parameters with the product's own mirror classes
(default init under the seed + oracle.fixtures.perturb_) and verify them against the
checksums the reference-side generator stored."""
import torch

from oracle import fixtures as FX


def build_fs_mirror(meta):
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    torch.manual_seed(meta["seed"])
    m = OnlineTransformerDADiarization(n_speakers=None, in_size=meta["in_size"], **meta["cfg"]).eval()
    FX.perturb_(m, meta["pseed"])
    FX.check_params(m.state_dict(), meta["checksums"])
    return m


def fs_kwargs(meta):
    c = meta["cfg"]
    return dict(n_heads=c["n_heads"], enc_n_layers=c["enc_n_layers"], dec_n_layers=c["dec_n_layers"],
                has_mask=c["has_mask"], mask_delay=c["mask_delay"])


def max_abs(a, b):
    return (a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max().item()


def build_ls_mirror(meta):
    from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization
    torch.manual_seed(meta["seed"])
    m = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=meta["in_size"], **meta["cfg"]).eval()
    FX.perturb_(m, meta["pseed"])
    FX.check_params(m.state_dict(), meta["checksums"])
    return m


def ls_kwargs(meta):
    c = meta["cfg"]
    return dict(n_heads=c["n_heads"], enc_n_layers=c["enc_n_layers"], dec_n_layers=c["dec_n_layers"],
                chunk=c["recurrent_chunk_size"], conv_delay=c["conv_delay"])
