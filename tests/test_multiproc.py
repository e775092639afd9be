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


def _dp_worker(rank, world, port, q):
    """One data-parallel training step on CPU: gradients of this rank's utterance shard from the oracle, the PRODUCT's
    flat layout + mean all-reduce (gloo), the oracle's Adam on the averaged flat gradient."""
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    from fs_eend_amd.shard import all_reduce_mean, flat_layout, shard_range
    from oracle import fixtures as FX
    from oracle import train_ref as TR
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = dict(n_units=256, n_heads=4, enc_n_layers=1, dec_n_layers=1, dropout=0.0, has_mask=True, max_seqlen=500,
               dec_dim_feedforward=256, mask_delay=0)
    torch.manual_seed(5)
    m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, **cfg)
    lens, nspk = [60, 60, 60, 60], [2, 3, 2, 1]
    feats, labels = FX.make_src(lens, 345, 1), FX.make_labels(lens, nspk, 2)
    a, b = shard_range(len(lens), rank, world)
    tr = TR.TrainRef(m.state_dict(), cfg, warmup=25, clip=5.0)
    shapes = [(k, tuple(tr.sd[k].shape)) for k in tr.pnames]
    offsets, total = flat_layout(shapes)

    def flat_grads(lo, hi):
        leaves = {k: tr.sd[k].clone().requires_grad_(True) for k in tr.pnames}
        sd = dict(tr.sd)
        sd.update(leaves)
        tot, *_ = TR.train_loss(sd, feats[lo:hi], labels[lo:hi], cfg, None, {})
        gs = torch.autograd.grad(tot, [leaves[k] for k in tr.pnames], allow_unused=True)
        buf = torch.zeros(total)
        for k, g in zip(tr.pnames, gs):
            if g is not None:
                buf[offsets[k]:offsets[k] + g.numel()] = g.flatten()
        return buf

    mine = flat_grads(a, b)
    reduced = all_reduce_mean(mine.clone())
    # what DDP computes: the mean of the per-rank gradients (each rank normalises its own loss, local BN statistics)
    want = sum(flat_grads(*shard_range(len(lens), r, world)) for r in range(world)) / world
    # Adam on the averaged gradient (same arithmetic on every rank -> bit-identical parameters)
    p = torch.cat([tr.sd[k].flatten() for k in tr.pnames])
    gcat = torch.cat([reduced[offsets[k]:offsets[k] + tr.sd[k].numel()] for k in tr.pnames])
    lr = TR.noam_lr(1, 256, 25)
    mom, var = 0.1 * gcat, 0.02 * gcat * gcat
    p_new = p - (lr / 0.1) * (mom / (var.sqrt() / (0.02 ** 0.5) + 1e-9))
    gathered = [None] * world
    dist.all_gather_object(gathered, p_new)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, float((reduced - want).abs().max()), float(want.abs().max()), bool(torch.equal(gathered[0], gathered[1])),
           float((mine - want).abs().max())))


def test_two_rank_data_parallel_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, scale, same, local_diff in res:
        assert err <= 1e-7 * max(scale, 1.0), (rank, err)        # all-reduced buffer == mean of the shard gradients
        assert same                                               # both ranks end with bit-identical parameters
        assert local_diff > 1e-4 * scale                          # ...although their local gradients differed


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


def _run_py(code, cwd="/tmp"):
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=cwd)


def test_dropin_import_paths():
    """The reference's import paths resolve to the HIP-backed classes once the launcher has arranged sys.path
    (fs_eend_amd.dropin.arrange_sys_path, what `python -m fs_eend_amd.run <script>` does).
    Here the "script" lives in an empty directory: the shim trees must also work stand-alone."""
    pre = "import sys, os; sys.path.insert(0, r'%s'); from fs_eend_amd.dropin import arrange_sys_path; " % ROOT
    fs = pre + "arrange_sys_path('/tmp/nowhere/FS-EEND/x.py'); "
    out = _run_py(fs +
                  "from nnet.model.onl_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import OnlineTransformerDADiarization as A; "
                  "from nnet.model.streaming_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import StreamingTransformerEDADiarization as S; "
                  "from nnet.utils.copy_params import copy_params_from_masked_to_streaming as C; "
                  "from nnet.modules.merge_tfm_encoder import TransformerEncoderFusionLayer; "
                  "import fs_eend_amd.fs_model as F; assert A is F.OnlineTransformerDADiarization; print('ok')")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr
    ls = pre + "arrange_sys_path('/tmp/nowhere/LS-EEND/x.py'); "
    out = _run_py(ls +
                  "from nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_emb_loss_mask "
                  "import OnlineConformerRetentionDADiarization as A, StreamingConv1d; "
                  "from nnet.conformer.encoder import ConformerEncoder; from nnet.modules.retention import MultiScaleRetention; "
                  "import fs_eend_amd.ls_model as L; assert A is L.OnlineConformerRetentionDADiarization; print('ok')")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr
    # output-side post-processing (train/utils/make_rttm.py, train/utils/loss.py DER report), input-side features
    for pre_ in (fs, ls):
        out = _run_py(pre_ +
                      "from train.utils.make_rttm import make_rttm; from train.utils.loss import calc_diarization_error, "
                      "report_diarization_error, batch_pit_n_speaker_loss, pit_loss_multispk, pad_labels, standard_loss; "
                      "from datasets.feature import extract_fbank, splice, subsample; import fs_eend_amd.feature as FE; "
                      "assert extract_fbank is FE.extract_fbank; import fs_eend_amd.postproc as P; "
                      "assert make_rttm is P.make_rttm and calc_diarization_error is P.calc_diarization_error; print('ok')")
        assert out.returncode == 0 and "ok" in out.stdout, out.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/FS-EEND"), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("flavour,script,model_mod,kept", [
    ("FS-EEND", "streaming_infer_dia.py", "nnet.model.onl_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm",
     ["train.oln_tfm_enc_dec", "train.oln_tfm_enc_dec_spk_pit", "datasets.diarization_dataset", "datasets.kaldi_data",
      "utlis.scheduler", "nnet.model.offl_tfm_enc_lstm_enc_dec"]),
    ("LS-EEND", "streaming_infer_dia.py",
     "nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_emb_loss_mask",
     ["train.oln_tfm_enc_dec", "train.oln_tfm_enc_dec_spk_pit_on_the_fly", "datasets.diarization_dataset_on_the_fly",
      "data_loaders.utils.my_distributed_sampler", "nnet.conformer.attention"])])
def test_dropin_binds_next_to_the_reference(flavour, script, model_mod, kept):
    """With the reference's project root present (the launcher's situation): the hot-path modules resolve to THIS
    repository, everything the shims do not provide still resolves to the reference's own files, and the modules the
    shims overlay (train/utils/loss.py) expose the reference's names next to the accelerated ones."""
    ref = f"/root/reference/{flavour}"
    code = ("import sys, os, importlib.util; sys.dont_write_bytecode = True; sys.path.insert(0, r'%s'); "
            "from fs_eend_amd.dropin import arrange_sys_path; arrange_sys_path(r'%s'); "
            "m = importlib.util.find_spec('%s'); assert m.origin.startswith(r'%s'), m.origin; "
            "bad = [n for n in %r if not importlib.util.find_spec(n).origin.startswith(r'%s')]; assert not bad, bad; "
            "sp = importlib.util.find_spec('train.utils.loss'); assert sp.origin.startswith(r'%s'), sp.origin; "
            "print('ok')") % (ROOT, os.path.join(ref, script), model_mod, ROOT, kept, ref, ROOT)
    out = _run_py(code)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr + out.stdout
    # the overlay: reference definitions + accelerated overrides in one module (torchmetrics is stubbed here only
    # because this container lacks it; the reference environment has it)
    code = ("import sys, types; sys.dont_write_bytecode = True; sys.path.insert(0, r'%s'); "
            "tm = types.ModuleType('torchmetrics'); tm.PermutationInvariantTraining = object; sys.modules['torchmetrics'] = tm; "
            "from fs_eend_amd.dropin import arrange_sys_path; arrange_sys_path(r'%s'); "
            "import train.utils.loss as L; import fs_eend_amd.postproc as P, fs_eend_amd.pit as PIT; "
            "assert L._REFERENCE_FILE.startswith(r'%s'), L._REFERENCE_FILE; "
            "assert L.standard_loss.__code__.co_filename.startswith(r'%s'); assert hasattr(L, 'batch_pit_loss'); "
            "assert L.calc_diarization_error is P.calc_diarization_error and L.batch_pit_n_speaker_loss is PIT.batch_pit_n_speaker_loss; "
            "print('ok')") % (ROOT, os.path.join(ref, script), ref, ref)
    out = _run_py(code)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr + out.stdout


def _bn_worker(rank, world, port, q):
    """SyncBatchNorm exchange of the LS-EEND training step on CPU tensors over gloo: the PRODUCT's host-side exchange
    (fs_eend_amd.shard.gather_bn_stats / all_reduce_bn_sums) around the oracle's restatement of the device arithmetic."""
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from fs_eend_amd.shard import BN_STATS, all_reduce_bn_sums, gather_bn_stats
    from oracle import bn_sync_ref as BR
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100)
    rows = [700, 1300]                                   # ranks own different frame counts (ragged batches)
    c_all = [torch.randn(r, 256, generator=g) * (1 + i) + 0.3 * i for i, r in enumerate(rows)]
    ds_all = [torch.randn(r, 256, generator=g) * 1e-3 for r in rows]
    gamma, beta = 1 + 0.2 * torch.randn(256, generator=g).double(), 0.1 * torch.randn(256, generator=g).double()
    c, ds = c_all[rank], ds_all[rank]
    st = BR.local_stats(c).float()
    assert st.numel() == BN_STATS
    table, R = gather_bn_stats(st, None)
    mean, var, n, var_unb = BR.merge(table)
    sums = BR.bwd_sums(ds, c, mean, var, gamma, beta).float()
    local_sums = sums.clone()
    all_reduce_bn_sums(sums, None)
    dc = BR.bwd_apply(ds, c, mean, var, gamma, beta, sums.double(), n)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, R, mean.numpy(), var.numpy(), var_unb.numpy(), n, dc.numpy(), local_sums.numpy()))     # plain arrays: no shared-memory handles


def test_two_rank_sync_batchnorm_exchange_equals_global_batch():
    """world_size 2: the merged statistics equal those of the concatenated batch, and every rank's input gradient equals
    the autograd gradient of swish(BatchNorm(c)) over the concatenated batch (torch.nn.SyncBatchNorm semantics); the
    weight / bias gradients stay rank-local sums (DDP averages them afterwards)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(100)
    rows = [700, 1300]
    c_all = [torch.randn(r, 256, generator=g) * (1 + i) + 0.3 * i for i, r in enumerate(rows)]
    ds_all = [torch.randn(r, 256, generator=g) * 1e-3 for r in rows]
    gamma, beta = 1 + 0.2 * torch.randn(256, generator=g).double(), 0.1 * torch.randn(256, generator=g).double()
    cat = torch.cat(c_all).double().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    mu, var = cat.mean(0), cat.var(0, unbiased=False)
    y = gm * (cat - mu) / torch.sqrt(var + 1e-5) + bt
    out = y * torch.sigmoid(y)
    out.backward(torch.cat(ds_all).double())
    off = 0
    res = [(r[0], r[1]) + tuple(torch.from_numpy(x) if hasattr(x, "dtype") else x for x in r[2:]) for r in res]
    for rank, R, mean, v, v_unb, n, dc, local_sums in res:
        assert R == 2 and n == sum(rows)
        assert torch.allclose(mean, mu.detach(), atol=1e-6) and torch.allclose(v, var.detach(), rtol=1e-5, atol=1e-7)
        assert torch.allclose(v_unb, cat.detach().var(0, unbiased=True), rtol=1e-5)
        assert torch.allclose(dc, cat.grad[off:off + rows[rank]], rtol=1e-4, atol=1e-9)
        off += rows[rank]
    # local sums of the two ranks add up to the BatchNorm weight / bias gradients of the global batch
    tot = res[0][7].double() + res[1][7].double()
    assert torch.allclose(tot[:256], bt.grad, rtol=1e-4, atol=1e-8) and torch.allclose(tot[256:], gm.grad, rtol=1e-4, atol=1e-8)
