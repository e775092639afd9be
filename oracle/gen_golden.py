"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py fs
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py ls

Imports the reference's own ``nnet`` package (FS-EEND and LS-EEND both call it ``nnet``,
hence one process per flavour), runs it on CPU fp32 with seeded weights/inputs and writes
inputs-seed + expected outputs + parameter checksums to tests/golden/*.npz.  Only data is
written: no reference source or bytecode enters the repository.
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import fixtures as FX

REF = "/root/reference"
ROWS = 16          # emb / attractor rows are stored subsampled (every ROWS-th frame)
ONLY = set(sys.argv[2:])       # optional: generate only the named cases (the others are reproducible bit for bit anyway)

FS_FULL = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, has_mask=True,
               max_seqlen=500, dec_dim_feedforward=2048, mask_delay=0)


def fs_cfg(**kw):
    c = dict(FS_FULL)
    c.update(kw)
    return c


FS_CASES = [
    # BASELINE config 1/2: conf/spk_onl_tfm_enc_dec_nonautoreg.yaml, T=500, max_nspks = max_speakers+2
    dict(name="fs_full_T500_c6", cfg=fs_cfg(), lengths=[500, 463], C=6, seed=0, pseed=11, xseed=777),
    dict(name="fs_full_T500_c4", cfg=fs_cfg(), lengths=[500], C=4, seed=0, pseed=11, xseed=778),
    # ragged, tiny, T < 64
    dict(name="fs_small_ragged", cfg=fs_cfg(enc_n_layers=2, dec_n_layers=1, dec_dim_feedforward=256),
         lengths=[37, 20, 64], C=4, seed=1, pseed=12, xseed=779),
    # Tp = 192 (odd multiple of 64), C = 3
    dict(name="fs_T130_c3", cfg=fs_cfg(enc_n_layers=1, dec_n_layers=2, dec_dim_feedforward=512),
         lengths=[130, 129], C=3, seed=2, pseed=13, xseed=780),
    # look-ahead in the attention masks
    dict(name="fs_delay2", cfg=fs_cfg(enc_n_layers=1, dec_n_layers=1, dec_dim_feedforward=256, mask_delay=2),
         lengths=[150, 90], C=5, seed=3, pseed=14, xseed=781),
    # encoder without mask (decoder always masks, model :116)
    dict(name="fs_nomask", cfg=fs_cfg(enc_n_layers=1, dec_n_layers=1, dec_dim_feedforward=256, has_mask=False),
         lengths=[100, 100], C=4, seed=4, pseed=15, xseed=782),
    # 10 speaker slots, single frame-ish edge: T = 1 and T = 65
    dict(name="fs_c10_T65", cfg=fs_cfg(enc_n_layers=1, dec_n_layers=1, dec_dim_feedforward=256),
         lengths=[65, 1], C=10, seed=5, pseed=16, xseed=783),
    # 12 speaker slots = max_speakers 10 + 2 (the dihard configs of LS-EEND; the largest slot count the kernels dispatch),
    # full-size model, ragged
    dict(name="fs_c12_T500", cfg=fs_cfg(), lengths=[500, 317], C=12, seed=8, pseed=19, xseed=787),
    # whole recordings at test time (the infer configs set chunk_size to the recording length): windows of more than 512 frames,
    # three groups of the grouped attention form with a ragged last one (Tp = 1344), full-size model
    dict(name="fs_long_T1300_c4", cfg=fs_cfg(), lengths=[1300, 900], C=4, seed=9, pseed=20, xseed=788),
    # ... and with look-ahead in the masks (keys of the next group in reach of a group's last queries), two groups (Tp = 704)
    dict(name="fs_long_delay2_T700", cfg=fs_cfg(enc_n_layers=1, dec_n_layers=1, dec_dim_feedforward=256, mask_delay=2),
         lengths=[700, 520], C=5, seed=10, pseed=21, xseed=789),
]

FS_FWD_CASES = [
    dict(name="fs_fwd_train", cfg=fs_cfg(enc_n_layers=2, dec_n_layers=2, dec_dim_feedforward=512),
         lengths=[200, 170], ncols=[4, 3], seed=6, pseed=17, xseed=784, lseed=785),
]

FS_STREAM_CASES = [
    dict(name="fs_stream_T60", cfg=fs_cfg(enc_n_layers=2, dec_n_layers=2, dec_dim_feedforward=2048),
         T=60, C=6, seed=7, pseed=18, xseed=786),
]


def _np(t):
    return t.detach().cpu().numpy()


def gen_fs():
    sys.path.insert(0, os.path.join(REF, "FS-EEND"))
    from nnet.model.onl_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import OnlineTransformerDADiarization
    from nnet.model.streaming_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import \
        StreamingTransformerEDADiarization
    from nnet.utils.copy_params import copy_params_from_masked_to_streaming
    from oracle import fs_eend_ref as R

    def build(case):
        torch.manual_seed(case["seed"])
        m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, **case["cfg"]).eval()
        FX.perturb_(m, case["pseed"])
        return m

    for case in FS_CASES:
        if ONLY and case["name"] not in ONLY:
            continue
        m = build(case)
        src = FX.make_src(case["lengths"], 345, case["xseed"])
        with torch.no_grad():
            logits, emb, attr = m.test(src, case["lengths"], case["C"])
            mine = R.fs_test(src, case["lengths"], m.state_dict(), n_heads=4,
                             enc_n_layers=case["cfg"]["enc_n_layers"], dec_n_layers=case["cfg"]["dec_n_layers"],
                             max_nspks=case["C"], has_mask=case["cfg"]["has_mask"],
                             mask_delay=case["cfg"]["mask_delay"])
        err = max((a - b).abs().max().item() for a, b in zip(logits, mine[0]))
        arrays = {}
        for i, (l, e, a) in enumerate(zip(logits, emb, attr)):
            arrays[f"logits{i}"] = _np(l)
            arrays[f"emb{i}"] = _np(e[::ROWS])
            arrays[f"attr{i}"] = _np(a[::ROWS])
        meta = dict(kind="fs_test", cfg=case["cfg"], lengths=case["lengths"], C=case["C"], seed=case["seed"],
                    pseed=case["pseed"], xseed=case["xseed"], rows=ROWS, in_size=345,
                    checksums=FX.param_checksums(m.state_dict()), torch=torch.__version__,
                    oracle_vs_reference_max_abs=err)
        p = FX.save_case(case["name"], meta, arrays)
        print(f"{case['name']}: oracle-vs-reference max|d logits| = {err:.2e} -> {os.path.relpath(p)}"
              f" ({os.path.getsize(p) / 1024:.0f} KiB)")
        assert err < 2e-6

    for case in FS_FWD_CASES:
        if ONLY and case["name"] not in ONLY:
            continue
        m = build(case)
        src = FX.make_src(case["lengths"], 345, case["xseed"])
        tgt = FX.make_labels(case["lengths"], case["ncols"], case["lseed"])
        with torch.no_grad():
            logits, loss, emb, attr = m(src, tgt, case["lengths"])
        arrays = {"emb_loss": _np(loss).reshape(1)}
        for i, (l, e, a) in enumerate(zip(logits, emb, attr)):
            arrays[f"logits{i}"] = _np(l)
            arrays[f"emb{i}"] = _np(e[::ROWS])
            arrays[f"attr{i}"] = _np(a[::ROWS])
        meta = dict(kind="fs_forward", cfg=case["cfg"], lengths=case["lengths"], ncols=case["ncols"],
                    seed=case["seed"], pseed=case["pseed"], xseed=case["xseed"], lseed=case["lseed"], rows=ROWS,
                    in_size=345, checksums=FX.param_checksums(m.state_dict()), torch=torch.__version__)
        p = FX.save_case(case["name"], meta, arrays)
        print(f"{case['name']}: emb_loss = {float(loss):.6f} -> {os.path.relpath(p)}")

    for case in FS_STREAM_CASES:
        if ONLY and case["name"] not in ONLY:
            continue
        m = build(case)
        cfg = dict(case["cfg"])
        sm = StreamingTransformerEDADiarization(in_size=345, **cfg).eval()
        copy_params_from_masked_to_streaming(m, sm)
        src = FX.make_src([case["T"]], 345, case["xseed"])[0]
        ys = []
        with torch.no_grad():
            batch = m.test([src], [case["T"]], case["C"])[0][0]
            for t in range(case["T"]):
                y = sm.test(src[t].view(1, 1, -1), case["C"])
                if y is not None:
                    ys.append(y)
            for _ in range(m.delay):
                y = sm.test(src[0].view(1, 1, -1), case["C"], dummy_conv_input=True)
                if y is not None:
                    ys.append(y)
        ys = torch.cat(ys, dim=1)[0]
        d = (ys - batch).abs().max().item()
        meta = dict(kind="fs_stream", cfg=case["cfg"], T=case["T"], C=case["C"], seed=case["seed"],
                    pseed=case["pseed"], xseed=case["xseed"], in_size=345,
                    checksums=FX.param_checksums(m.state_dict()), torch=torch.__version__,
                    stream_vs_batch_max_abs=d)
        p = FX.save_case(case["name"], meta, {"stream_logits": _np(ys), "batch_logits": _np(batch)})
        print(f"{case['name']}: reference streaming vs batch max|d| = {d:.2e} -> {os.path.relpath(p)}")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "fs"
    if which == "fs":
        gen_fs()
    elif which == "ls":
        from oracle.gen_golden_ls import gen_ls
        gen_ls()
    else:
        raise SystemExit("usage: gen_golden.py fs|ls")
