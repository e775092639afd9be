"""GPU: one whole FS-EEND training step in HIP (fs_eend_amd.train.FsTrainStep through the SpeakerDiarization surface)
against the golden vectors produced by the reference's own training_step / standard_loss / Adam / NoamScheduler
(oracle/gen_golden_train.py), plus size-independent properties at BASELINE config 4's full size.

Bars (VERDICT r01 / north_star): loss within 1e-4; per-parameter gradient norms within 1e-2 relative (plus a small
absolute floor for parameters whose gradient is ~0); single gradient entries within ENTRY_BAR of the tensor's gradient
norm; parameters after Adam within a fraction of the learning rate.
"""
import math

import numpy as np
import pytest
import torch

from oracle import fixtures as FX
from tests.helpers import build_fs_mirror

pytestmark = pytest.mark.gpu
CASES = FX.list_cases("fs_train_")
ENTRY_BAR = 5e-2        # |entry error| / ||gradient of that tensor||: bf16 gradient operands, a handful of entries per tensor


def _slice_index(numel, n=24):
    a = np.arange(min(12, numel))
    b = (np.arange(12) * 7919 + 13) % numel
    return np.concatenate([a, b]).astype(np.int64)[:n]


def _module(meta, dev):
    from fs_eend_amd.trainer import SpeakerDiarization
    m = build_fs_mirror(meta).to(dev).train()
    hp = dict(data=dict(max_speakers=4, label_delay=0), training=dict(lr=1.0, warm_steps=meta["warm"], schedule_scale=1.0,
                                                                      grad_clip=meta["clip"], batch_size=len(meta["lengths"])))
    return SpeakerDiarization(hp, m, {}, dict(lr=1.0, betas=(0.9, 0.98), eps=1e-9), dict(warmup_steps=meta["warm"], scale=1.0),
                              None, pit=meta["pit"]), m


@pytest.mark.parametrize("force_gacc_stream", [False, True])
@pytest.mark.parametrize("name", CASES)
def test_train_step_vs_reference(hip_lib, dev, name, force_gacc_stream, monkeypatch):
    if force_gacc_stream:
        # the packed-stream data-gradient GEMM of the attention in-projections (gemm_acc_stream.hip + eend_layernorm_bwd_f32 in place of the
        # tiled GEMM with the LayerNorm backward in its epilogue) is taken from 96 k rows; the goldens are smaller: same bars with it forced on
        from fs_eend_amd.train import TrainStepBase
        monkeypatch.setattr(TrainStepBase, "gacc_stream_min_rows", 0)
    meta, arr = FX.load_case(name)
    mod, m = _module(meta, dev)
    feats = [f.to(dev) for f in FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])]
    labels = [l.to(dev) for l in FX.make_labels(meta["lengths"], meta["nspk"], meta["lseed"])]
    names = meta["param_names"]
    report = []
    for s in range(meta["steps"]):
        loss = mod.training_step([feats, labels, None], s)
        mod.backward()
        eng = mod._engine()
        torch.cuda.synchronize()
        want = arr[f"s{s}_loss"]
        got = (float(loss), float(mod.logged["train/pit_loss"]), float(mod.logged["train/emb_loss"]))
        print(f"{name} step {s}: loss {got[0]:.6f} (ref {want[0]:.6f})  bce {got[1]:.6f} ({want[1]:.6f})  emb {got[2]:.6f} ({want[2]:.6f})")
        assert abs(got[1] - want[1]) < 1e-4 and abs(got[2] - want[2]) < 1e-4 and abs(got[0] - want[0]) < 1e-4
        if s == 0:
            tot = arr["s0_gradnorm"][0]
            worst = 0.0
            for i, k in enumerate(names):
                g = eng.flat.g(k)
                if k in meta["nograd"]:
                    assert float(g.abs().max()) == 0.0, k
                    continue
                gn = float(g.double().norm())
                ref = arr["grad_norms"][i]
                err = abs(gn - ref) / max(ref, 1e-3 * tot)
                worst = max(worst, err)
                report.append((err, k, gn, ref))
                idx = _slice_index(g.numel())
                sl = g.flatten()[torch.as_tensor(idx, device=dev)].cpu().numpy()
                serr = np.abs(sl - arr["grad_slices"][i][:len(idx)]).max() / max(ref, 1e-3 * tot)
                report.append((serr, k + " [entries]", float(np.abs(sl).max()), float(np.abs(arr["grad_slices"][i]).max())))
            report.sort(reverse=True)
            for err, k, a, b in report[:14]:
                print(f"   {err:.3e}  {k}: {a:.4e} vs {b:.4e}")
            bad = [(e, k) for e, k, _, _ in report if e > 1e-2 and not k.endswith("[entries]")]
            assert not bad, bad[:10]
            bad = [(e, k) for e, k, _, _ in report if e > ENTRY_BAR and k.endswith("[entries]")]
            assert not bad, bad[:10]
        lr = mod.optimizer_step()
        torch.cuda.synchronize()
        assert abs(lr - arr[f"s{s}_lr"][0]) < 1e-9 * max(1.0, arr[f"s{s}_lr"][0]) + 1e-12
        gn_got = float(eng.gsumsq.sqrt())
        assert abs(gn_got - arr[f"s{s}_gradnorm"][0]) < 1e-2 * arr[f"s{s}_gradnorm"][0], (gn_got, arr[f"s{s}_gradnorm"][0])
        # parameters after Adam: an entry moves by ~lr per step (Adam normalises), so compare in units of lr
        n_bad = n_all = 0
        for i, k in enumerate(names):
            p = eng.flat.p(k)
            idx = _slice_index(p.numel())
            got_p = p.flatten()[torch.as_tensor(idx, device=dev)].cpu().numpy()
            d = np.abs(got_p - arr[f"s{s}_param_slices"][i][:len(idx)])
            n_bad += int((d > 0.15 * lr * (s + 1) + 1e-6).sum())
            n_all += len(idx)
            assert d.max() < 2.5 * lr * (s + 1) + 1e-6, (k, d.max(), lr)
        print(f"   step {s}: {n_bad} of {n_all} sampled parameter entries more than 0.15 lr off the reference's")
        assert n_bad <= 0.03 * n_all, (n_bad, n_all)
        bn = m.enc.bn
        assert np.abs(bn.running_mean.cpu().numpy() - arr[f"s{s}_bn_mean"]).max() < 1e-4
        assert np.abs(bn.running_var.cpu().numpy() - arr[f"s{s}_bn_var"]).max() < 1e-3


def test_autograd_dropin_matches_reference(hip_lib, dev):
    """The reference's own training recipe on the mirror model: model(feats, labels, ilens) with autograd enabled,
    torch standard_loss + emb_loss, loss.backward(), clip_grad_norm_, torch.optim.Adam -- the backward that runs is
    the HIP one.  Same golden bars as the native step."""
    from fs_eend_amd.trainer import prepare_labels
    from oracle import train_ref as TR
    meta, arr = FX.load_case("fs_train_small")
    m = build_fs_mirror(meta).to(dev).train()
    feats = [f.to(dev) for f in FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])]
    raw = [l.to(dev) for l in FX.make_labels(meta["lengths"], meta["nspk"], meta["lseed"])]
    opt = torch.optim.Adam(m.parameters(), lr=1.0, betas=(0.9, 0.98), eps=1e-9)
    for s in range(2):
        labels = prepare_labels(raw, meta["lengths"])
        opt.zero_grad()
        preds, emb_loss, embs, attrs = m(feats, labels, meta["lengths"])
        loss = TR.standard_loss(preds, labels) + emb_loss
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), meta["clip"])
        want = arr[f"s{s}_loss"]
        assert abs(float(loss) - want[0]) < 1e-4, (float(loss), want[0])
        assert abs(float(gn) - arr[f"s{s}_gradnorm"][0]) < 1e-2 * arr[f"s{s}_gradnorm"][0]
        if s == 0:
            tot = arr["s0_gradnorm"][0]
            for i, (k, p) in enumerate(m.named_parameters()):
                if k in meta["nograd"]:
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0
                    continue
                ref = arr["grad_norms"][i]
                assert abs(float(p.grad.double().norm()) - ref) / max(ref, 1e-3 * tot) < 1e-2, k
        for gq in opt.param_groups:
            gq["lr"] = float(arr[f"s{s}_lr"][0])
        opt.step()
    with pytest.raises(Exception):
        m.eval()
        m(feats, labels, meta["lengths"])                 # gradient-enabled eval-mode call: refused loudly


@pytest.mark.parametrize("p_drop", [0.1, 0.3])
def test_train_step_with_dropout_vs_oracle(hip_lib, dev, p_drop):
    """Dropout on (the reference's yaml trains with it): the kernels' masks are a hash of (seed, element index), so the
    oracle driven with the SAME masks (oracle/dropout_ref.py) must give the same loss and gradients -- all ten sites of
    both layer types, forward and backward, two consecutive forwards (the mask changes with the forward count).  The
    reference's own Philox masks are not reproducible; p = 0 is pinned by the golden cases above."""
    from fs_eend_amd import ops
    from fs_eend_amd.train import FsTrainStep
    from fs_eend_amd.trainer import prepare_labels
    from oracle import dropout_ref as DR
    from oracle import train_ref as TR
    meta, _ = FX.load_case("fs_train_small")
    meta = dict(meta, cfg=dict(meta["cfg"], dropout=p_drop))
    m = build_fs_mirror(meta).to(dev).train()
    sd = {k: v.detach().cpu().double() if v.is_floating_point() else v.cpu() for k, v in m.state_dict().items()}
    eng = FsTrainStep(m, warmup=meta["warm"], grad_clip=meta["clip"], drop_seed=1234)
    assert eng.drop_p == p_drop
    eng.prep_weights()
    feats = [f.to(dev) for f in FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])]
    raw = [l.to(dev) for l in FX.make_labels(meta["lengths"], meta["nspk"], meta["lseed"])]
    labels = prepare_labels(raw, meta["lengths"])
    cfg = dict(meta["cfg"], n_units=256)
    pn = [k for k, v in sd.items() if v.is_floating_point() and v.dim() >= 1
          and not k.endswith(("running_mean", "running_var", "pos_enc.pe"))]
    losses = []
    for fwd in (1, 2):
        bf = eng.forward(feats, labels, meta["lengths"])
        eng.backward(bf)
        torch.cuda.synchronize()
        Tp = ops.frames_pad(max(meta["lengths"]))
        drop = DR.HashDropout(p_drop, 1234, fwd, Tp)
        leaves = {k: sd[k].clone().requires_grad_(True) for k in pn}
        sdd = dict(sd)
        sdd.update(leaves)
        tot, bce, emb, _, _ = TR.train_loss(sdd, [f.cpu().double() for f in feats], [l.cpu().double() for l in raw], cfg,
                                            dtype=torch.float64, drop=drop)
        grads = dict(zip(pn, torch.autograd.grad(tot, [leaves[k] for k in pn], allow_unused=True)))
        got = (float(bf.loss[0]), float(bf.loss[1]))
        print(f"p={p_drop} fwd {fwd}: bce {got[0]:.6f} (oracle {float(bce):.6f})  emb {got[1]:.6f} ({float(emb):.6f})")
        assert abs(got[0] - float(bce)) < 2e-4 and abs(got[1] - float(emb)) < 1e-4
        losses.append(got[0])
        totn = math.sqrt(sum(float((g ** 2).sum()) for g in grads.values() if g is not None))
        worst = []
        for k in pn:
            g = eng.flat.g(k)
            if grads[k] is None:
                assert float(g.abs().max()) == 0.0, k
                continue
            ref = grads[k]
            err = float((g.detach().cpu().double() - ref).norm()) / max(float(ref.norm()), 1e-3 * totn)
            worst.append((err, k))
        worst.sort(reverse=True)
        print("   worst rel. gradient errors:", [(f"{e:.2e}", k) for e, k in worst[:5]])
        assert worst[0][0] < 2e-2, worst[:5]          # whole-tensor relative L2 error (bf16 gradient operands)
    assert losses[0] != losses[1]                     # a new mask every forward
    # dropout off on request (validation-style forward through the training engine) == the p = 0 path
    bf = eng.forward(feats, labels, meta["lengths"], dropout=False)
    torch.cuda.synchronize()
    tot0, bce0, emb0, _, _ = TR.train_loss(sd, [f.cpu().double() for f in feats], [l.cpu().double() for l in raw], cfg,
                                           dtype=torch.float64)
    assert abs(float(bf.loss[0]) - float(bce0)) < 1e-4


def test_dropout_mask_statistics(hip_lib, dev):
    """The hash masks behave like Bernoulli(1-p) draws: keep rate, scaling, independence between sites and steps
    (measured on the FFN hidden activation through the C-ABI entry point that applies them)."""
    import ctypes
    from fs_eend_amd import lib as L
    from fs_eend_amd.train import _call, drop_site_seed, drop_step_seed
    M, N, K = 4096, 1024, 256
    a = torch.zeros(M, K, dtype=torch.float16, device=dev)
    w = torch.zeros(N, K, dtype=torch.float16, device=dev)
    bias = torch.ones(N, dtype=torch.float32, device=dev)            # relu(0 + 1) = 1 everywhere -> the output IS the mask
    outs = []
    for fwd, site in ((1, 4), (1, 20), (2, 4)):
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        spec = L.Dropout(drop_site_seed(drop_step_seed(9, fwd), site), int(round(0.1 * (1 << 24))), 1.0 / 0.9)
        _call("eend_linear_relu_train_f16", a, K, w, K, bias, out, N, M, N, K, ctypes.byref(spec))
        outs.append(out.float())
    torch.cuda.synchronize()
    for o in outs:
        kept = o != 0
        assert abs(float(kept.float().mean()) - 0.9) < 2e-3
        assert float((o[kept] - 1.0 / 0.9).abs().max()) < 1e-3
        assert abs(float(kept.float().mean(0).std()) - math.sqrt(0.09 / M)) < 1e-3     # per-column rates spread like binomial
        assert abs(float(kept.float().mean(1).std()) - math.sqrt(0.09 / N)) < 2e-3
    for i, j in ((0, 1), (0, 2)):                                    # different site / different step: independent masks
        both = ((outs[i] != 0) & (outs[j] != 0)).float().mean()
        assert abs(float(both) - 0.81) < 3e-3


def test_label_preparation_matches_oracle(dev):
    from fs_eend_amd.trainer import prepare_labels
    from oracle import train_ref as TR
    lens, nspk = [200, 170, 120, 33], [2, 3, 1, 4]
    raw = FX.make_labels(lens, nspk, 99)
    want = TR.prepare_labels(raw, lens)
    got = prepare_labels([r.to(dev) for r in raw], lens)
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)


def test_full_size_step_properties(hip_lib, dev):
    """BASELINE config 4 at full size (B = 64 utterances x T = 500, 4 speakers, shipped yaml): finite, deterministic,
    the loss goes down over a few steps, never-graded slices stay zero, utterance order does not matter."""
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    from fs_eend_amd.train import FsTrainStep, never_graded
    from fs_eend_amd.trainer import prepare_labels
    cfg = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.0, has_mask=True, max_seqlen=500,
               dec_dim_feedforward=2048, mask_delay=0)
    B = 64
    lens = [500] * B
    feats = [f.to(dev) for f in FX.make_src(lens, 345, 777)]
    raw = [l.to(dev) for l in FX.make_labels(lens, [4] * B, 778)]
    labels = prepare_labels(raw, lens)

    def run(order, steps):
        torch.manual_seed(0)
        m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, **cfg).to(dev).train()
        eng = FsTrainStep(m, warmup=400, grad_clip=5.0)        # lr = 7.8e-6 * step: small enough that repeating the batch must lower the loss
        losses = []
        for _ in range(steps):
            out = eng.step([feats[i] for i in order], [labels[i] for i in order], [lens[i] for i in order])
            losses.append(float(out["loss"]))
        torch.cuda.synchronize()
        return eng, losses

    eng, losses = run(list(range(B)), 4)
    print("full-size losses:", losses, "peak HBM GB:", torch.cuda.max_memory_allocated() / 2 ** 30)
    assert all(math.isfinite(l) for l in losses) and losses[-1] < losses[0]
    assert torch.isfinite(eng.flat.params).all() and torch.isfinite(eng.flat.grads).all()
    for k in eng.flat.names:
        if never_graded(k):
            assert float(eng.flat.g(k).abs().max()) == 0.0
    eng2, losses2 = run(list(range(B)), 1)
    eng3, losses3 = run(list(reversed(range(B))), 1)
    assert losses2[0] == losses[0]                                  # bit-reproducible
    assert abs(losses3[0] - losses[0]) < 1e-5                       # batch order only changes summation order
    rel = float((eng3.flat.grads - eng2.flat.grads).norm() / eng2.flat.grads.norm())
    assert rel < 2e-3, rel
