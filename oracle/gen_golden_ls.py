"""LS-EEND half of the golden-vector generator (see oracle/gen_golden.py; build container only)."""
import os
import sys

import numpy as np
import torch

from oracle import fixtures as FX

REF = "/root/reference"
ROWS = 16

ONLY = set(sys.argv[2:])       # optional: generate only the named cases

LS_FULL = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, max_seqlen=1000,
               recurrent_chunk_size=500, feed_forward_expansion_factor=4, dec_dim_feedforward=2048,
               conv_expansion_factor=2, conv_kernel_size=16, half_step_residual=True, conv_delay=9)


def ls_cfg(**kw):
    c = dict(LS_FULL)
    c.update(kw)
    return c


LS_CASES = [
    # BASELINE config 3: T=2000 = 4 chunks of 500, max_nspks = max_speakers + 2 = 10
    dict(name="ls_full_T2000_c10", cfg=ls_cfg(), lengths=[2000], C=10, seed=20, pseed=31, xseed=801),
    # T % 500 != 0 (zero padding to the chunk multiple, look-ahead conv sees zeros), ragged batch
    dict(name="ls_T1234_ragged", cfg=ls_cfg(enc_n_layers=2, dec_n_layers=1, dec_dim_feedforward=512),
         lengths=[1234, 777], C=4, seed=21, pseed=32, xseed=802),
    # the reference's own self-test sizes are D=64; ours must keep D=256/H=4, so: tiny chunk instead
    dict(name="ls_chunk10_T30", cfg=ls_cfg(enc_n_layers=2, dec_n_layers=2, dec_dim_feedforward=256,
                                           recurrent_chunk_size=10, conv_kernel_size=7),
         lengths=[30, 17], C=3, seed=22, pseed=33, xseed=803),
    dict(name="ls_chunk64_T200", cfg=ls_cfg(enc_n_layers=1, dec_n_layers=1, dec_dim_feedforward=256,
                                            recurrent_chunk_size=64),
         lengths=[200], C=6, seed=23, pseed=34, xseed=804),
    dict(name="ls_T500_c3", cfg=ls_cfg(enc_n_layers=1, dec_n_layers=1, dec_dim_feedforward=256),
         lengths=[500, 499], C=3, seed=24, pseed=35, xseed=805),
    # 12 slots: conf/spk_onl_conformer_retention_enc_dec_nonautoreg_dihard{2,3}{,_infer}.yaml set max_speakers 10 -> max_nspks 12
    # (train/oln_tfm_enc_dec.py:35,186); full-size model, two chunks
    dict(name="ls_c12_T1000", cfg=ls_cfg(), lengths=[1000], C=12, seed=27, pseed=38, xseed=809),
    # round 5: further random initialisations at the largest slot counts (the margin of the one C = 12 seed was 10 %)
    dict(name="ls_c12_T1000_s2", cfg=ls_cfg(), lengths=[1000], C=12, seed=41, pseed=42, xseed=811),
    dict(name="ls_c12_T1000_s3", cfg=ls_cfg(), lengths=[1000], C=12, seed=43, pseed=44, xseed=812),
    dict(name="ls_c10_T1500_s2", cfg=ls_cfg(), lengths=[1500, 1100], C=10, seed=45, pseed=46, xseed=813),
]

LS_FWD_CASES = [
    dict(name="ls_fwd_train", cfg=ls_cfg(enc_n_layers=2, dec_n_layers=2, dec_dim_feedforward=512,
                                         recurrent_chunk_size=100),
         lengths=[300, 250], ncols=[6, 4], seed=25, pseed=36, xseed=806, lseed=807),
]

LS_STREAM_CASES = [
    dict(name="ls_stream_T120", cfg=ls_cfg(enc_n_layers=2, dec_n_layers=2, dec_dim_feedforward=2048,
                                           recurrent_chunk_size=50),
         T=120, C=10, seed=26, pseed=37, xseed=808),
]


def _np(t):
    return t.detach().cpu().numpy()


def gen_ls():
    sys.path.insert(0, os.path.join(REF, "LS-EEND"))
    from nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_emb_loss_mask \
        import OnlineConformerRetentionDADiarization, StreamingConv1d
    from oracle import ls_eend_ref as R

    def build(case):
        torch.manual_seed(case["seed"])
        m = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **case["cfg"]).eval()
        FX.perturb_(m, case["pseed"])
        return m

    def okw(cfg):
        return dict(n_heads=cfg["n_heads"], enc_n_layers=cfg["enc_n_layers"], dec_n_layers=cfg["dec_n_layers"],
                    chunk=cfg["recurrent_chunk_size"], conv_delay=cfg["conv_delay"])

    for case in LS_CASES:
        if ONLY and case["name"] not in ONLY:
            continue
        m = build(case)
        src = FX.make_src(case["lengths"], 345, case["xseed"])
        with torch.no_grad():
            logits, emb, attr = m.test(src, case["lengths"], case["C"])
            mine = R.ls_test(src, case["lengths"], m.state_dict(), max_nspks=case["C"], **okw(case["cfg"]))
        err = max((a - b).abs().max().item() for a, b in zip(logits, mine[0]))
        arrays = {}
        for i, (l, e, a) in enumerate(zip(logits, emb, attr)):
            arrays[f"logits{i}"] = _np(l)
            arrays[f"emb{i}"] = _np(e[::ROWS])
            arrays[f"attr{i}"] = _np(a[::ROWS])
        meta = dict(kind="ls_test", cfg=case["cfg"], lengths=case["lengths"], C=case["C"], seed=case["seed"],
                    pseed=case["pseed"], xseed=case["xseed"], rows=ROWS, in_size=345,
                    checksums=FX.param_checksums(m.state_dict()), torch=torch.__version__,
                    oracle_vs_reference_max_abs=err)
        p = FX.save_case(case["name"], meta, arrays)
        print(f"{case['name']}: oracle-vs-reference max|d logits| = {err:.2e} -> {os.path.relpath(p)}"
              f" ({os.path.getsize(p) / 1024:.0f} KiB)")
        assert err < 5e-6

    for case in LS_FWD_CASES:
        if ONLY and case["name"] not in ONLY:
            continue
        m = build(case)
        src = FX.make_src(case["lengths"], 345, case["xseed"])
        tgt = FX.make_labels(case["lengths"], case["ncols"], case["lseed"])
        with torch.no_grad():
            logits, loss, emb, attr = m(src, tgt, case["lengths"])
            mine = R.ls_forward(src, tgt, case["lengths"], m.state_dict(), **okw(case["cfg"]))
        err = max((a - b).abs().max().item() for a, b in zip(logits, mine[0]))
        arrays = {"emb_loss": _np(loss).reshape(1)}
        for i, (l, e, a) in enumerate(zip(logits, emb, attr)):
            arrays[f"logits{i}"] = _np(l)
            arrays[f"emb{i}"] = _np(e[::ROWS])
            arrays[f"attr{i}"] = _np(a[::ROWS])
        meta = dict(kind="ls_forward", cfg=case["cfg"], lengths=case["lengths"], ncols=case["ncols"],
                    seed=case["seed"], pseed=case["pseed"], xseed=case["xseed"], lseed=case["lseed"], rows=ROWS,
                    in_size=345, checksums=FX.param_checksums(m.state_dict()), torch=torch.__version__)
        p = FX.save_case(case["name"], meta, arrays)
        print(f"{case['name']}: emb_loss = {float(loss):.6f} (oracle {float(mine[1]):.6f}), "
              f"oracle-vs-reference logits {err:.2e} -> {os.path.relpath(p)}")
        assert err < 5e-6 and abs(float(loss) - float(mine[1])) < 1e-6

    # streaming: drive the reference exactly as LS-EEND/streaming_infer_dia.py:52-97 does
    for case in LS_STREAM_CASES:
        if ONLY and case["name"] not in ONLY:
            continue
        m = build(case)
        src = FX.make_src([case["T"]], 345, case["xseed"])[0]
        C = case["C"]
        scnn = StreamingConv1d(m.n_units, m.n_units, kernel_size=2 * m.delay + 1).eval()
        scnn.conv.load_state_dict(m.cnn.state_dict())
        n_enc, n_dec = len(m.enc.encoder.layers), len(m.dec.layers)
        ret_states = [dict() for _ in range(n_enc)]
        caches = [torch.zeros(1, m.n_units, m.enc.encoder._conv_kernel_size - 1) for _ in range(n_enc)]
        dec_states = [dict() for _ in range(n_dec)]
        preds, dec_t = [], 0

        def step(emb_t, dec_t):
            e = scnn(emb_t.transpose(1, 2))
            if e is None:
                return None, dec_t
            e = e.transpose(1, 2)
            e = e / torch.norm(e, dim=-1, keepdim=True)
            a = m.dec.forward_one_step(e, dec_t, C, dec_states)
            a = a / torch.norm(a, dim=-1, keepdim=True)
            return torch.matmul(e.unsqueeze(-2), a.transpose(-1, -2)).squeeze(-2), dec_t + 1

        with torch.no_grad():
            batch = m.test([src], [case["T"]], C)[0][0]
            for t in range(case["T"]):
                e = m.enc.forward_one_step(src[t:t + 1].unsqueeze(0), t, ret_states, caches)
                y, dec_t = step(e, dec_t)
                if y is not None:
                    preds.append(y)
            for _ in range(m.delay):
                y, dec_t = step(torch.zeros(1, 1, m.n_units), dec_t)
                if y is not None:
                    preds.append(y)
        ys = torch.cat(preds, dim=1)[0]
        s = R.LsStreamingRef(m.state_dict(), n_heads=4, enc_n_layers=n_enc, dec_n_layers=n_dec,
                             conv_kernel_size=case["cfg"]["conv_kernel_size"])
        mine = []
        with torch.no_grad():
            for t in range(case["T"]):
                y = s.step(src[t].view(1, 1, -1), C)
                if y is not None:
                    mine.append(y)
            for _ in range(m.delay):
                y = s.step(None, C)
                if y is not None:
                    mine.append(y)
        mine = torch.cat(mine, dim=1)[0]
        err = (mine - ys).abs().max().item()
        d = (ys - batch).abs().max().item()
        meta = dict(kind="ls_stream", cfg=case["cfg"], T=case["T"], C=C, seed=case["seed"], pseed=case["pseed"],
                    xseed=case["xseed"], in_size=345, checksums=FX.param_checksums(m.state_dict()),
                    torch=torch.__version__, stream_vs_batch_max_abs=d, oracle_vs_reference_max_abs=err)
        p = FX.save_case(case["name"], meta, {"stream_logits": _np(ys), "batch_logits": _np(batch)})
        print(f"{case['name']}: oracle-vs-reference streaming {err:.2e}; reference streaming vs batch {d:.2e}"
              f" -> {os.path.relpath(p)}")
        assert err < 5e-6
