"""GPU: one whole LS-EEND training step in HIP (fs_eend_amd.train_ls.LsTrainStep through the SpeakerDiarization
surface) against the golden vectors produced by the reference's own LS training_step / standard_loss /
pit_loss_multispk / Adam / NoamScheduler (oracle/gen_golden_train_ls.py), the retention backward kernels alone against
an fp64 restatement, and size-independent properties at BASELINE config 4's LS size (T = 1000).

Bars: loss within 1e-4; per-parameter gradient norms within 1e-2 relative (plus a small absolute
floor for parameters whose gradient is ~0); single gradient entries within ENTRY_BAR of the tensor's gradient norm;
parameters after Adam within a fraction of the learning rate; BatchNorm running statistics within 1e-4 / 1e-3.

Conditioned bar: no hand-picked constant above 1e-2.  A few gradient tensors -- the retention q / k
projections behind a per-head LayerNorm at its eps floor -- are ill-conditioned at random init: the reference's own fp32
step and the same step in fp64 disagree on them ten to ninety times more than on the median tensor (stored per tensor in
the golden as `grad_gap_l2` by oracle/gen_golden_train_ls.py; measured on the fly with the fp32 / fp64 oracle in the dropout
test).  The bar of a tensor is max(1e-2, COND_K * gap), COND_K = 2^13 = the ratio of the unit roundoffs of f16 (the forward
operand type of the HIP path; its gradient operands are bf16, 2^16) and f32 (the reference's): what the reference's own
discrepancy on that tensor becomes at the HIP path's working precision.
"""
import math

import numpy as np
import pytest
import torch

from oracle import fixtures as FX
from tests.helpers import build_ls_mirror

pytestmark = pytest.mark.gpu
CASES = FX.list_cases("ls_train_")
ENTRY_BAR = 5e-2        # |entry error| / ||gradient of that tensor||: bf16 gradient operands, a handful of entries per tensor
COND_K = 8192.0         # 2^13 = eps(f16) / eps(f32): see the module docstring


def cond_bar(gap):
    """Relative error allowed on a gradient tensor whose fp32-vs-fp64 discrepancy in the reference is `gap`."""
    return max(1e-2, COND_K * float(gap))


def _slice_index(numel, n=24):
    a = np.arange(min(12, numel))
    b = (np.arange(12) * 7919 + 13) % numel
    return np.concatenate([a, b]).astype(np.int64)[:n]


def _module(meta, dev):
    from fs_eend_amd.trainer import SpeakerDiarization
    m = build_ls_mirror(meta).to(dev).train()
    hp = dict(data=dict(max_speakers=8, label_delay=0), training=dict(lr=1.0, warm_steps=meta["warm"], schedule_scale=1.0,
                                                                      grad_clip=meta["clip"], batch_size=len(meta["lengths"])))
    return SpeakerDiarization(hp, m, {}, dict(lr=1.0, betas=(0.9, 0.98), eps=1e-9), dict(warmup_steps=meta["warm"], scale=1.0),
                              None, pit=meta["pit"]), m


@pytest.mark.parametrize("force_proj_stream", [False, True])
@pytest.mark.parametrize("name", CASES)
def test_ls_train_step_vs_reference(hip_lib, dev, name, force_proj_stream, monkeypatch):
    if force_proj_stream:
        # the packed-stream retention projection (proj_stream.hip: one pass for the forward's f16 operands and the backward's bf16 head
        # rows) is taken from 48 k rows; the goldens are smaller, so the same bars are checked with it forced on
        from fs_eend_amd.train_ls import LsTrainStep
        monkeypatch.setattr(LsTrainStep, "proj_stream_min_rows", 0)
        monkeypatch.setattr(LsTrainStep, "gacc_stream_min_rows", 0)          # ... and the packed-stream data-gradient GEMM (gemm_acc_stream.hip)
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
        # first step: the 1e-4 bar.  Later steps start from parameters that already differ from the reference's by a fraction
        # of lr per entry (Adam divides by sqrt(v): entries whose gradient is near zero move by +-lr whatever its rounding; the
        # parameter check below bounds that), so their losses get that much slack
        ltol = 1e-4 + (0.5 * arr[f"s{s - 1}_lr"][0] if s else 0.0)
        assert abs(got[1] - want[1]) < ltol and abs(got[2] - want[2]) < ltol and abs(got[0] - want[0]) < ltol
        if s == 0:
            tot = arr["s0_gradnorm"][0]
            gaps = arr["grad_gap_l2"]                    # the reference's own fp32-vs-fp64 discrepancy per tensor
            entries = []
            for i, k in enumerate(names):
                g = eng.flat.g(k)
                if k in meta["nograd"]:
                    assert float(g.abs().max()) == 0.0, k
                    continue
                assert bool(torch.isfinite(g).all()), k
                gn = float(g.double().norm())
                ref = arr["grad_norms"][i]
                report.append((abs(gn - ref) / max(ref, 1e-3 * tot), k, gn, ref, cond_bar(gaps[i])))
                idx = _slice_index(g.numel())
                sl = g.flatten()[torch.as_tensor(idx, device=dev)].cpu().numpy()
                entries.append((np.abs(sl - arr["grad_slices"][i][:len(idx)]).max() / max(ref, 1e-3 * tot), k))
            report.sort(reverse=True)
            entries.sort(reverse=True)
            for err, k, a, b, bar in report[:12]:
                print(f"   {err:.3e}  (bar {bar:.1e})  {k}: {a:.4e} vs {b:.4e}")
            print("   worst single entries:", [(f"{e:.2e}", k) for e, k in entries[:4]])
            bad = [(e, k, bar) for e, k, _, _, bar in report if e > bar]
            assert not bad, bad[:10]
            assert entries[0][0] < ENTRY_BAR, entries[:5]
        lr = mod.optimizer_step()
        torch.cuda.synchronize()
        assert abs(lr - arr[f"s{s}_lr"][0]) < 1e-9 * max(1.0, arr[f"s{s}_lr"][0]) + 1e-12
        gn_got = float(eng.gsumsq.sqrt())
        assert abs(gn_got - arr[f"s{s}_gradnorm"][0]) < 1e-2 * arr[f"s{s}_gradnorm"][0], (gn_got, arr[f"s{s}_gradnorm"][0])
        n_bad = n_all = 0
        for i, k in enumerate(names):
            p = eng.flat.p(k)
            idx = _slice_index(p.numel())
            got_p = p.flatten()[torch.as_tensor(idx, device=dev)].cpu().numpy()
            d = np.abs(got_p - arr[f"s{s}_param_slices"][i][:len(idx)])
            n_bad += int((d > 0.15 * lr * (s + 1) + 1e-6).sum())
            n_all += len(idx)
            assert d.max() < 2.5 * lr * (s + 1) + 1e-6, (k, d.max(), lr)
        assert n_bad <= 0.03 * n_all, (n_bad, n_all)
        sd = m.state_dict()
        for j, k in enumerate(meta["bn_keys"]):                       # conv-module BatchNorm running statistics
            # from the second step on the parameters themselves differ by a fraction of lr (see above), and so do the statistics
            tol = (1e-4 if k.endswith("running_mean") else 1e-3) + 0.5 * lr * s
            assert np.abs(sd[k].cpu().numpy() - arr[f"s{s}_bn"][j]).max() < tol, (k, s)


@pytest.mark.parametrize("p_drop", [0.0, 0.1, 0.3])
def test_ls_train_step_with_dropout_vs_oracle(hip_lib, dev, p_drop):
    """Dropout on (the shipped LS yaml trains with 0.1): the kernels' masks are a hash of (seed, element index), so the
    oracle driven with the SAME masks (oracle/dropout_ref.HashDropout through ls_eend_ref's `drop` hook) must give the
    same loss and gradients -- the six sites of every Conformer block (feed_forward.py:51,53, attention.py:112,
    convolution.py:148) and the five of every decoder layer (merge_retnet_layer.py:82,298,307,311,312), forward and
    backward, two consecutive forwards.  The reference's own Philox masks cannot be reproduced; p = 0 is pinned above."""
    from fs_eend_amd import ops
    from fs_eend_amd.train_ls import LsTrainStep
    from fs_eend_amd.trainer import prepare_labels
    from oracle import dropout_ref as DR
    from oracle import train_ls_ref as TL
    meta, _ = FX.load_case("ls_train_small")
    meta = dict(meta, cfg=dict(meta["cfg"], dropout=p_drop))
    m = build_ls_mirror(meta).to(dev).train()
    sd = {k: v.detach().cpu().double() if v.is_floating_point() else v.cpu() for k, v in m.state_dict().items()}
    eng = LsTrainStep(m, warmup=meta["warm"], grad_clip=meta["clip"], drop_seed=4321)
    assert eng.drop_p == p_drop
    eng.prep_weights()
    feats = [f.to(dev) for f in FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])]
    raw = [l.to(dev) for l in FX.make_labels(meta["lengths"], meta["nspk"], meta["lseed"])]
    labels = prepare_labels(raw, meta["lengths"])
    cfg = dict(meta["cfg"], n_units=256)
    L = cfg["recurrent_chunk_size"]
    Tp = ops.frames_pad(math.ceil(max(meta["lengths"]) / L) * L)
    pn = [k for k, v in sd.items() if v.is_floating_point() and v.dim() >= 1
          and not k.endswith(("running_mean", "running_var", "pos_enc.pe", ".angle", ".decay"))]
    losses = []
    for fwd in (1, 2):
        bf = eng.forward(feats, labels, meta["lengths"])
        eng.backward(bf)
        torch.cuda.synchronize()
        drop = DR.HashDropout(p_drop, 4321, fwd, Tp)
        leaves = {k: sd[k].clone().requires_grad_(True) for k in pn}
        sdd = dict(sd)
        sdd.update(leaves)
        tot, bce, emb, _, _ = TL.train_loss(sdd, [f.cpu().double() for f in feats], [l.cpu().double() for l in raw], cfg,
                                            dtype=torch.float64, drop=drop)
        grads = dict(zip(pn, torch.autograd.grad(tot, [leaves[k] for k in pn], allow_unused=True)))
        # the same step with the same masks in fp32: the oracle's own discrepancy per tensor (conditioning, module docstring)
        leaves32 = {k: sd[k].float().clone().requires_grad_(True) for k in pn}
        sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
        sd32.update(leaves32)
        tot32 = TL.train_loss(sd32, [f.cpu().float() for f in feats], [l.cpu().float() for l in raw], cfg, dtype=torch.float32,
                              drop=DR.HashDropout(p_drop, 4321, fwd, Tp))[0]
        grads32 = dict(zip(pn, torch.autograd.grad(tot32, [leaves32[k] for k in pn], allow_unused=True)))
        got = (float(bf.loss[0]), float(bf.loss[1]))
        print(f"LS p={p_drop} fwd {fwd}: bce {got[0]:.6f} (oracle {float(bce):.6f})  emb {got[1]:.6f} ({float(emb):.6f})")
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
            den = max(float(ref.norm()), 1e-3 * totn)
            err = float((g.detach().cpu().double() - ref).norm()) / den
            gap = float((grads32[k].double() - ref).norm()) / den
            worst.append((err, k, cond_bar(gap)))
        worst.sort(reverse=True)
        print("   worst rel. gradient errors:", [(f"{e:.2e} (bar {b:.1e})", k) for e, k, b in worst[:5]])
        # whole-tensor relative L2 error against the fp64 oracle (bf16 gradient operands), bar = max(1e-2, COND_K * the oracle's own
        # fp32-vs-fp64 discrepancy on that tensor).  Measured on MI355X (round 3 - 5): every tensor <= 6e-3 except the first encoder
        # block's retention q / k projections (the per-head LayerNorm backward in front of them cancels most of its input): 1.1e-2
        # at p = 0, 1.3e-2 at the shipped p = 0.1, 2.6e-2 at the stress value 0.3 -- the tensors with the largest gaps (3 - 4e-6).
        assert all(e < b for e, k, b in worst), [w for w in worst if w[0] >= w[2]][:5]
    assert p_drop == 0.0 or losses[0] != losses[1]    # a new mask every forward
    bf = eng.forward(feats, labels, meta["lengths"], dropout=False)
    torch.cuda.synchronize()
    tot0, bce0, emb0, _, _ = TL.train_loss(sd, [f.cpu().double() for f in feats], [l.cpu().double() for l in raw], cfg,
                                           dtype=torch.float64)
    assert abs(float(bf.loss[0]) - float(bce0)) < 1e-4 and abs(float(bf.loss[1]) - float(emb0)) < 1e-4


def _ret_reference(q, k, v, o, L):
    """fp64: dq, dk, dv of out_t = q_t . sum_{s <= t, chunk-wise + prefix} k_s (x) v_s contracted with o (the detached
    scales already folded into o).  q, k, v, o: (N, H, T, 64)."""
    T = q.shape[2]
    causal = torch.tril(torch.ones(T, T, dtype=torch.float64, device=q.device))
    A = (o @ v.transpose(-1, -2)) * causal
    S = (q @ k.transpose(-1, -2)) * causal
    return A @ k, A.transpose(-1, -2) @ q, S.transpose(-1, -2) @ o


@pytest.mark.parametrize("nseq,Tv,L", [(3, 1000, 500), (2, 300, 100), (5, 512, 64), (2, 500, 500)])
def test_retention_core_backward_kernels(hip_lib, dev, nseq, Tv, L):
    """eend_retention_bwd_bf16 against the closed form, with gate = identity-like inputs folded out: drive the entry with
    rhat = 0 (so d_g = 0 and the per-head LayerNorm backward reduces to d_r = rc * (d_rhat - mean)), compare
    dq / sk*dk / dv with the fp64 products of the bf16-rounded operands.  Covers chunk boundaries that are not multiples
    of the 64 / 128-row tiles (L = 500, 100), several chunks (prefix / suffix states) and the single-chunk case."""
    from fs_eend_amd import ops
    from fs_eend_amd.train import _call
    H, D = 4, 256
    Tp = ops.frames_pad(Tv)
    M = nseq * Tp
    g = torch.Generator(device="cpu").manual_seed(nseq * 1000 + Tv)
    def heads(scale):                                    # (nseq, H, Tp, 64) bf16 + its transposed copy
        x = (torch.randn(nseq, H, Tp, 64, generator=g) * scale).to(torch.bfloat16)
        return x.to(dev).contiguous(), x.transpose(-1, -2).contiguous().to(dev)
    q, qt = heads(0.5)
    k, kt = heads(0.5)
    v, vt = heads(1.0)
    dctx = (torch.randn(M, D, generator=g) * 1e-3).to(dev)
    gate = (torch.randn(M, D, generator=g)).to(torch.float16).to(dev)
    rhat = torch.zeros(M, D, dtype=torch.float16, device=dev)
    rc = (0.5 + torch.rand(M, H, generator=g)).to(dev)
    nc = Tv // L
    ot = torch.empty(M * D, dtype=torch.bfloat16, device=dev)
    ott = torch.empty(M * D, dtype=torch.bfloat16, device=dev)
    kv_ws = torch.empty(nseq * H * nc * 4096, dtype=torch.float32, device=dev)
    g_ws = torch.empty(nseq * H * nc * 4096, dtype=torch.float32, device=dev)
    st = torch.empty(nseq * H * nc * 6 * 4096, dtype=torch.bfloat16, device=dev)
    dq = torch.full((M, 4 * D), float("nan"), dtype=torch.bfloat16, device=dev)
    _call("eend_retention_bwd_bf16", q, qt, k, kt, v, vt, dctx, gate, D, rhat, rc, ot, ott, kv_ws, g_ws, st, dq, 4 * D, nseq, H, Tp, L, Tv, 0.125)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dq.float()).all())
    # chunk lengths up to 512 (round 6): the [d][t] copies and the transposed o~ scratch are not read -- NULL is accepted, same bits
    dq2 = torch.full((M, 4 * D), float("nan"), dtype=torch.bfloat16, device=dev)
    _call("eend_retention_bwd_bf16", q, None, k, None, v, None, dctx, gate, D, rhat, rc, ot, None, kv_ws, g_ws, st, dq2, 4 * D, nseq, H, Tp, L, Tv, 0.125)
    torch.cuda.synchronize()
    assert torch.equal(dq2, dq)
    # expected o~ (fp64 from the same inputs): d_rhat = dctx * swish(g); d_r = rc * (d_rhat - mean_head(d_rhat))  [rhat = 0]
    gg = gate.double()
    drh = dctx.double() * gg * torch.sigmoid(gg)
    drh = drh.view(M, H, 64)
    o_exp = (rc.double()[:, :, None] * (drh - drh.mean(-1, keepdim=True)))
    t_idx = torch.arange(M, device=dev) % Tp
    o_exp[t_idx >= Tv] = 0
    o_got = ot.view(M, H, 64).double()
    assert float((o_got - o_exp).abs().max()) <= 1e-2 * float(o_exp.abs().max()) + 1e-12
    assert float(dq[:, 3 * D:].float().abs().max()) == 0.0                     # d_g = dctx * rhat * swish'(g) = 0
    # retention products from the kernel's own (bf16) o~
    o4 = o_got.view(nseq, Tp, H, 64).permute(0, 2, 1, 3)[:, :, :Tv]
    q4, k4, v4 = (x.double()[:, :, :Tv] for x in (q, k, v))
    dq_e, dk_e, dv_e = _ret_reference(q4, k4, v4, o4, L)
    got = dq.double().view(nseq, Tp, 4, H, 64).permute(2, 0, 3, 1, 4)        # (4, nseq, H, Tp, 64)
    for name, e, gt, sc in (("dq", dq_e, got[0], 1.0), ("dk", dk_e, got[1], 0.125), ("dv", dv_e, got[2], 1.0)):
        err = float((gt[:, :, :Tv] - sc * e).norm()) / float((sc * e).norm())
        worst = float((gt[:, :, :Tv] - sc * e).abs().max()) / float((sc * e).abs().max())
        print(f"retention bwd nseq={nseq} Tv={Tv} L={L}: {name} rel L2 {err:.2e}, worst entry {worst:.2e}")
        assert err < 6e-3 and worst < 3e-2, (name, err, worst)
        assert float(gt[:, :, Tv:].abs().max()) == 0.0 if Tp > Tv else True   # slab padding rows: zero gradients


def test_ls_full_size_step_properties(hip_lib, dev):
    """BASELINE config 4, LS half, at full size (B = 64 utterances x T = 1000 = two retention chunks, 4 speakers, shipped
    yaml shapes): finite, bit-reproducible, never-graded slices stay zero, repeating the batch lowers the loss."""
    from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization
    from fs_eend_amd.train_ls import LsTrainStep, never_graded
    from fs_eend_amd.trainer import prepare_labels
    cfg = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.0, max_seqlen=1000, recurrent_chunk_size=500,
               feed_forward_expansion_factor=4, dec_dim_feedforward=2048, conv_expansion_factor=2, conv_kernel_size=16,
               half_step_residual=True, conv_delay=9)
    B = 64
    lens = [1000] * B
    feats = [f.to(dev) for f in FX.make_src(lens, 345, 777)]
    raw = [l.to(dev) for l in FX.make_labels(lens, [4] * B, 778)]
    labels = prepare_labels(raw, lens)

    def run(steps):
        torch.manual_seed(0)
        m = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **cfg).to(dev).train()
        eng = LsTrainStep(m, warmup=400, grad_clip=5.0)
        losses = []
        for _ in range(steps):
            out = eng.step(feats, labels, lens)
            losses.append(float(out["loss"]))
        torch.cuda.synchronize()
        return eng, losses

    e1, l1 = run(4)
    assert all(math.isfinite(x) for x in l1), l1
    assert l1[-1] < l1[0], l1
    assert bool(torch.isfinite(e1.flat.params).all()) and bool(torch.isfinite(e1.flat.grads).all())
    for k in e1.flat.names:
        if never_graded(k):
            assert float(e1.flat.g(k).abs().max()) == 0.0, k
    e2, l2 = run(4)
    assert l1 == l2 and torch.equal(e1.flat.params, e2.flat.params)          # fixed summation order: bit-reproducible
    print("LS full-size losses:", [f"{x:.5f}" for x in l1], " peak HBM GB:", torch.cuda.max_memory_allocated() / 2 ** 30)


def test_ls_autograd_dropin_matches_reference(hip_lib, dev):
    """The reference's own LS training recipe on the mirror model: model(feats, labels, ilens) with autograd enabled,
    torch standard_loss + emb_loss, loss.backward(), clip_grad_norm_ -- the backward that runs is the HIP one."""
    from fs_eend_amd.trainer import prepare_labels
    from oracle import train_ref as TR
    meta, arr = FX.load_case("ls_train_small")
    m = build_ls_mirror(meta).to(dev).train()
    feats = [f.to(dev) for f in FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])]
    raw = [l.to(dev) for l in FX.make_labels(meta["lengths"], meta["nspk"], meta["lseed"])]
    labels = prepare_labels(raw, meta["lengths"])
    preds, emb_loss, embs, attrs = m(feats, labels, meta["lengths"])
    loss = TR.standard_loss(preds, labels) + emb_loss
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(m.parameters(), meta["clip"])
    want = arr["s0_loss"]
    assert abs(float(loss) - want[0]) < 1e-4, (float(loss), want[0])
    assert abs(float(gn) - arr["s0_gradnorm"][0]) < 1e-2 * arr["s0_gradnorm"][0]
    tot = arr["s0_gradnorm"][0]
    for i, (k, p) in enumerate(m.named_parameters()):
        if k in meta["nograd"]:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
            continue
        ref = arr["grad_norms"][i]
        assert abs(float(p.grad.double().norm()) - ref) / max(ref, 1e-3 * tot) < 1e-2, k
    with pytest.raises(Exception):
        m.eval()
        m(feats, labels, meta["lengths"])                 # gradient-enabled eval-mode call: refused loudly
