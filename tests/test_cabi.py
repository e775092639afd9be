"""CPU: the C-ABI shared library builds for gfx950, loads, and exports exactly the entry
points include/eend_hip.h declares; the product path refuses to run without the GPU
(no CPU / oracle fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_protos():
    txt = open(os.path.join(ROOT, "include", "eend_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(eend_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        protos[m.group(1)] = n
    return protos


def test_header_and_binding_agree(hip_lib):
    from fs_eend_amd import lib as L
    protos = _header_protos()
    assert len(protos) >= 11
    assert set(protos) == set(L.PROTOTYPES), set(protos) ^ set(L.PROTOTYPES)
    for name, n in protos.items():
        assert len(L.PROTOTYPES[name]) == n, f"{name}: header has {n} args, binding {len(L.PROTOTYPES[name])}"


def test_library_exports_every_symbol(hip_lib):
    from fs_eend_amd import lib as L
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in _header_protos():
        assert hasattr(raw, name), f"libeend_hip.so does not export {name}"
    assert hip_lib.eend_abi_version() == 5


def test_no_torch_types_in_abi():
    txt = open(os.path.join(ROOT, "include", "eend_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)          # signatures only, not the citations
    assert "torch" not in txt and "at::" not in txt and "Tensor" not in txt


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "fs-eend_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "oracle/" not in src and "oracle." not in src, f"{f} reaches into oracle/"


def test_product_path_fails_loudly_without_gpu(hip_lib):
    """CPU tensors must raise, never silently fall back to eager / the oracle."""
    from fs_eend_amd import ops
    from fs_eend_amd.lib import EendHipError
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    a = torch.zeros(64, 64, dtype=torch.float16)
    with pytest.raises(EendHipError):
        ops.linear(a, a, None, a.clone())
    m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, n_units=256, n_heads=4, enc_n_layers=1,
                                       dec_n_layers=1, dropout=0.1, has_mask=True, max_seqlen=500,
                                       dec_dim_feedforward=64).eval()
    with pytest.raises(EendHipError):
        m.test([torch.zeros(10, 345)], [10], 4)


def test_missing_library_raises(monkeypatch, tmp_path):
    from fs_eend_amd import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.EendHipError):
        L.load()


def test_state_dict_contract():
    """Reference checkpoint compatibility: 106 entries, 9 974 450 parameters."""
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, n_units=256, n_heads=4, enc_n_layers=4,
                                       dec_n_layers=2, dropout=0.1, has_mask=True, max_seqlen=500,
                                       dec_dim_feedforward=2048)
    sd = m.state_dict()
    assert len(sd) == 106
    assert sum(p.numel() for p in m.parameters()) == 9974450
    for k in ("enc.transformer_encoder.layers.0.self_attn.in_proj_weight", "dec.attractor_decoder.layers.0.norm12.weight",
              "cnn.weight", "dec.pos_enc.pe", "dec.encoder.weight", "enc.bn.running_mean"):
        assert k in sd
    assert sd["dec.pos_enc.pe"].shape == (1, 5000, 256) and sd["cnn.weight"].shape == (256, 256, 19)


def test_packed_stream_shape_predicates(hip_lib):
    """The pure shape predicates / stream sizes of the round-4 packed-weight entries (host arithmetic only, no launch): each packed
    form covers a stated set of shapes and reports everything else as unsupported, so that the caller keeps the unpacked entry."""
    L = hip_lib
    # layer tail: F/32 pairs of 16-KB items (+ 8 out-projection items)
    assert L.eend_ffn_stream_elems(2048, 1) == (8 + 2 * 64) * 8192 and L.eend_ffn_stream_elems(1024, 0) == 2 * 32 * 8192
    # ... one launch addresses rows with 32-bit offsets: the C ABI serves larger M in ranges of this many rows (the plain
    # entry returned EEND_EINVAL beyond ~2.03 M rows); a multiple of both tile sizes, inside the kernel's own guard
    cap = L.eend_ffn_stream_max_rows(256)
    assert cap % 384 == 0 and 2_000_000 < cap and (cap + 65536) * 1024 < 2 ** 31 and L.eend_ffn_stream_max_rows(512) == cap
    assert L.eend_ffn_stream_max_rows(1024) % 384 == 0 and (L.eend_ffn_stream_max_rows(1024) + 65536) * 2048 < 2 ** 31
    # decoder layer head: 8 out-projection + 24 in-projection items; 1..12 slots, Tp a multiple of 64 / 32 / 16 frames (C <= 3 / 6 / 12)
    assert L.eend_spk_stream_elems() == 32 * 8192
    assert [c for c in range(0, 14) if L.eend_spk_stream_ok(c, 512)] == list(range(1, 13))
    assert L.eend_spk_stream_ok(6, 480) and not L.eend_spk_stream_ok(6, 500) and not L.eend_spk_stream_ok(3, 96) and L.eend_spk_stream_ok(10, 80)
    # time-axis attention: 4 heads x 6 items
    assert L.eend_inproj_attn_packed_elems() == 4 * 6 * 8192
    # look-ahead conv: 8 items per tap; 256 channels, up to 24 taps
    assert L.eend_conv_stream_elems(19) == 19 * 8 * 8192
    assert L.eend_conv_stream_ok(256, 19, 9) and not L.eend_conv_stream_ok(256, 25, 9) and not L.eend_conv_stream_ok(128, 19, 9)
    # encoder input: 320 < in_size <= 384, Tp a multiple of 16, padded weight rows
    assert L.eend_encoder_input_ok(345, 512, 384) and L.eend_encoder_input_ok(384, 64, 384)
    assert not L.eend_encoder_input_ok(320, 512, 320) and not L.eend_encoder_input_ok(345, 500, 384) and not L.eend_encoder_input_ok(345, 512, 345)
