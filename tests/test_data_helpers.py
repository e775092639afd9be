"""CPU: sampler / chunking / checkpoint-averaging / config helpers against fixtures produced by the
reference's own MyDistributedSampler class and chunk-grid functions (oracle/gen_golden_data.py)."""
import json
import os

import pytest
import torch

from fs_eend_amd import config as CFG
from fs_eend_amd import data as D
from fs_eend_amd.trainer import average_checkpoints

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_sampler.json")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD))


def test_sampler_pairs_match_reference(gold):
    for c in gold["sampler"]:
        seen = []
        for rank in range(c["world"]):
            s = D.MyDistributedSampler(list(range(c["n"])), num_replicas=c["world"], rank=rank, shuffle=c["shuffle"], seed=c["seed"],
                                       drop_last=c["drop_last"])
            s.set_epoch(c["epoch"])
            got = [[int(i), int(sd)] for i, sd in s]
            assert got == c["pairs"][rank], (c["n"], c["world"], c["epoch"], rank)
            assert len(got) == len(s)
            seen += [i for i, _ in got]
        if not c["drop_last"]:
            assert set(seen) == set(range(c["n"]))               # every item is visited by some rank


def test_chunk_grid_matches_reference(gold):
    for c in gold["grid"]:
        assert D.count_frames(c["data_len"], c["size"], c["step"]) == c["count"]
        got = [[a, b] for a, b in D.gen_frame_indices(c["data_len"], c["size"], c["step"], c["use_last_samples"], label_delay=c["label_delay"])]
        assert got == c["chunks"], c


def test_on_the_fly_chunk_matches_reference(gold):
    for c in gold["on_the_fly"]:
        rec, st, ed = D.on_the_fly_chunk(("r", c["data_len"], 0, c["data_len"]), c["seed"], c["chunk_size"], c["subsampling"])
        assert (st, ed) == (c["st"], c["ed"]), c
    assert D.on_the_fly_chunk(("r", 100, 10, 60), 1, 1000, 10, data_type="val") == ("r", 10, 60)


def test_chunk_table_and_checkpoint_selection():
    rows = D.chunk_table([("a", 125.0), ("b", 49.9)], chunk_size=500, chunk_step=500, frame_shift=80, rate=8000, subsampling=10,
                         use_last_samples=True)
    assert rows == [("a", 0, 5000), ("a", 5000, 10000), ("a", 10000, 12500), ("b", 0, 4990)]
    files = ["epoch=89-step=1.ckpt", "epoch=90-step=2.ckpt", "epoch=99-step=3.ckpt", "epoch=100-step=4.ckpt", "last.ckpt", "x.txt"]
    assert D.select_epoch_checkpoints(files, 90, 99) == ["epoch=90-step=2.ckpt", "epoch=99-step=3.ckpt"]


def test_average_checkpoints_matches_reference_arithmetic():
    g = torch.Generator().manual_seed(0)
    sds = [{"w": torch.randn(5, 3, generator=g), "n": torch.tensor(7 + i)} for i in range(3)]
    avg = average_checkpoints(sds)
    want_w = sds[0]["w"] / 3 + sds[1]["w"] / 3 + sds[2]["w"] / 3          # the reference accumulates param / len(ckpts)
    assert torch.equal(avg["w"], 0.0 + want_w)
    assert abs(float(avg["n"]) - 8.0) < 1e-6


def test_yaml_ref_loader():
    cfg = CFG.load(CFG.FS_EEND_SIMU)
    assert cfg["model"]["params"]["max_seqlen"] == 500 and isinstance(cfg["model"]["params"]["max_seqlen"], int)
    kw = CFG.model_kwargs(cfg)
    assert kw["in_size"] == 345 and kw["n_units"] == 256 and kw["dec_dim_feedforward"] == 2048 and kw["dropout"] == 0.1
    ls = CFG.load(CFG.LS_EEND_SIMU)
    assert ls["model"]["params"]["max_seqlen"] == 1000 and ls["model"]["params"]["recurrent_chunk_size"] == 500
    txt = "log:\n  model_name: m1\n  log_dir: !ref ./logs/<log[model_name]>/v<a[1]>\na: [3, 4]\nb: !ref <a[0]>\n"
    out = CFG.loads(txt)
    assert out["log"]["log_dir"] == "./logs/m1/v4" and out["b"] == 3


@pytest.mark.skipif(not os.path.isdir("/root/reference/FS-EEND/conf"), reason="needs the reference tree (build container only)")
def test_loader_reads_the_reference_configs():
    """Every shipped reference config parses, and the committed tables carry the reference's values."""
    import glob
    n = 0
    for path in glob.glob("/root/reference/*/conf/*.yaml"):
        cfg = CFG.load(path)
        assert isinstance(cfg, dict) and "model" in cfg
        n += 1
    assert n >= 10
    ref = CFG.load("/root/reference/FS-EEND/conf/spk_onl_tfm_enc_dec_nonautoreg.yaml")
    mine = CFG.load(CFG.FS_EEND_SIMU)
    assert ref["model"]["params"] == mine["model"]["params"]
    for k in ("max_speakers", "context_recp", "chunk_size", "subsampling", "label_delay", "feat_type"):
        assert ref["data"][k] == mine["data"][k]
    for k in ("lr", "grad_clip", "warm_steps", "schedule_scale", "batch_size"):
        assert ref["training"][k] == mine["training"][k]
    ref = CFG.load("/root/reference/LS-EEND/conf/spk_onl_conformer_retention_enc_dec_nonautoreg.yaml")
    mine = CFG.load(CFG.LS_EEND_SIMU)
    assert ref["model"]["params"] == mine["model"]["params"]
