"""CPU, world_size 2 over gloo: the N>1 path of bench.py / evaluation -- disjoint complete
utterance shards and the whole-job throughput reduction (sum of frames / max of time)."""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    from fs_eend_amd.shard import job_throughput, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_range(n_items, rank, world)
    frames, secs = (b - a) * 500.0, 1.0 + rank            # rank 1 is the slow one
    thr = job_throughput(frames, secs)
    gathered = [None] * world
    dist.all_gather_object(gathered, (a, b))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, a, b, thr, gathered))


@pytest.mark.parametrize("n_items", [7, 64])
def test_two_rank_sharding_and_throughput(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, a0, b0, t0, g0), (_, a1, b1, t1, _) = res
    assert a0 == 0 and b0 == a1 and b1 == n_items              # disjoint, complete, contiguous
    assert abs((b0 - a0) - (b1 - a1)) <= 1
    assert t0 == t1 == pytest.approx(n_items * 500.0 / 2.0)    # sum(frames) / max(time) on every rank
    assert g0 == [(a0, b0), (a1, b1)]


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    from fs_eend_amd.shard import shard_range
    for n in (0, 1, 5, 8, 63, 64, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_dropin_import_paths():
    """The reference's import paths resolve to the HIP-backed classes."""
    code = ("import sys; sys.path.insert(0, r'%s'); "
            "from nnet.model.onl_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import OnlineTransformerDADiarization as A; "
            "from nnet.model.streaming_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import StreamingTransformerEDADiarization as S; "
            "from nnet.utils.copy_params import copy_params_from_masked_to_streaming as C; "
            "from nnet.modules.merge_tfm_encoder import TransformerEncoderFusionLayer; "
            "import fs_eend_amd.fs_model as F; assert A is F.OnlineTransformerDADiarization; print('ok')")
    out = subprocess.run([sys.executable, "-c", code % os.path.join(ROOT, "FS-EEND")], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr
    code = ("import sys; sys.path.insert(0, r'%s'); "
            "from nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_emb_loss_mask "
            "import OnlineConformerRetentionDADiarization as A, StreamingConv1d; "
            "from nnet.conformer.encoder import ConformerEncoder; from nnet.modules.retention import MultiScaleRetention; "
            "import fs_eend_amd.ls_model as L; assert A is L.OnlineConformerRetentionDADiarization; print('ok')")
    out = subprocess.run([sys.executable, "-c", code % os.path.join(ROOT, "LS-EEND")], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr
    # output-side post-processing (train/utils/make_rttm.py, train/utils/loss.py DER report)
    for flavour in ("FS-EEND", "LS-EEND"):
        code = ("import sys; sys.path.insert(0, r'%s'); "
                "from train.utils.make_rttm import make_rttm; from train.utils.loss import calc_diarization_error, report_diarization_error, batch_pit_n_speaker_loss, pit_loss_multispk, pad_labels; "
                "from datasets.feature import extract_fbank, splice, subsample; import fs_eend_amd.feature as FE; assert extract_fbank is FE.extract_fbank; "
                "import fs_eend_amd.postproc as P; assert make_rttm is P.make_rttm and calc_diarization_error is P.calc_diarization_error; print('ok')")
        out = subprocess.run([sys.executable, "-c", code % os.path.join(ROOT, flavour)], capture_output=True, text=True, cwd="/tmp")
        assert out.returncode == 0 and "ok" in out.stdout, out.stderr
