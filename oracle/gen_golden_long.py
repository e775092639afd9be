"""Long-horizon golden vectors (BASELINE config 5: one hour of audio = 36 000 frames, retention state carried across
72 chunks; FS-EEND K/V-cache decode far beyond the training chunk) -- runs ONLY in the build container.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_long.py ls_batch | ls_stream | ls_stream64 | fs_stream

What runs is the imported reference (eval mode, CPU fp32):
  * ls_batch : OnlineConformerRetentionDADiarization.test on 1 x T = 36 000, max_nspks = 10 (LS model :125-147)
  * ls_stream: the frame-by-frame driver of LS-EEND/streaming_infer_dia.py:52-97 (enc.forward_one_step ->
               StreamingConv1d -> L2 -> dec.forward_one_step -> L2 -> dot) over the same hour, same weights
  * fs_stream: StreamingTransformerEDADiarization.test frame by frame for T = 5000 (FS streaming model :31-60,
               K/V caches grow with t) next to the masked model's batch test on the same frames
Only subsampled output rows are stored (the first and last 600 frames and every 37th in between).
"""
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import fixtures as FX

REF = "/root/reference"
LS_FULL = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, max_seqlen=1000,
               recurrent_chunk_size=500, feed_forward_expansion_factor=4, dec_dim_feedforward=2048,
               conv_expansion_factor=2, conv_kernel_size=16, half_step_residual=True, conv_delay=9)
FS_FULL = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, has_mask=True, max_seqlen=500,
               dec_dim_feedforward=2048, mask_delay=0)
LS_HOUR = dict(T=36000, C=10, seed=71, pseed=72, xseed=873)
FS_LONG = dict(T=5000, C=6, seed=73, pseed=74, xseed=875)


def row_index(T):
    mid = np.arange(600, max(600, T - 600), 37)
    return np.unique(np.concatenate([np.arange(min(600, T)), mid, np.arange(max(0, T - 600), T)])).astype(np.int64)


def build_ls():
    sys.path.insert(0, os.path.join(REF, "LS-EEND"))
    from nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_emb_loss_mask \
        import OnlineConformerRetentionDADiarization, StreamingConv1d
    torch.manual_seed(LS_HOUR["seed"])
    m = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **LS_FULL).eval()
    FX.perturb_(m, LS_HOUR["pseed"])
    return m, StreamingConv1d


def ls_meta(m, **extra):
    return dict(cfg=LS_FULL, lengths=[LS_HOUR["T"]], C=LS_HOUR["C"], seed=LS_HOUR["seed"], pseed=LS_HOUR["pseed"],
                xseed=LS_HOUR["xseed"], in_size=345, checksums=FX.param_checksums(m.state_dict()), torch=torch.__version__, **extra)


def gen_ls_batch():
    m, _ = build_ls()
    T, C = LS_HOUR["T"], LS_HOUR["C"]
    src = FX.make_src([T], 345, LS_HOUR["xseed"])
    t0 = time.time()
    with torch.no_grad():
        logits, emb, attr = m.test(src, [T], C)
    dt = time.time() - t0
    idx = row_index(T)
    arrays = dict(rows=idx, logits=logits[0][torch.as_tensor(idx)].numpy(), emb=emb[0][torch.as_tensor(idx[::8])].numpy())
    p = FX.save_case("ls_hour_c10", ls_meta(m, kind="ls_hour", reference_cpu_seconds=dt), arrays)
    print(f"ls_hour_c10: reference model.test on T={T}, C={C} took {dt:.1f} s; {len(idx)} rows -> {os.path.relpath(p)} "
          f"({os.path.getsize(p) / 1024:.0f} KiB)")


def gen_ls_stream():
    m, StreamingConv1d = build_ls()
    T, C = LS_HOUR["T"], LS_HOUR["C"]
    src = FX.make_src([T], 345, LS_HOUR["xseed"])[0]
    scnn = StreamingConv1d(m.n_units, m.n_units, kernel_size=2 * m.delay + 1).eval()
    scnn.conv.load_state_dict(m.cnn.state_dict())
    n_enc, n_dec = len(m.enc.encoder.layers), len(m.dec.layers)
    ret_states = [dict() for _ in range(n_enc)]
    caches = [torch.zeros(1, m.n_units, m.enc.encoder._conv_kernel_size - 1) for _ in range(n_enc)]
    dec_states = [dict() for _ in range(n_dec)]
    idx = set(row_index(T).tolist())
    kept, dec_t = {}, 0

    def step(emb_t, dec_t):
        e = scnn(emb_t.transpose(1, 2))
        if e is None:
            return None, dec_t
        e = e.transpose(1, 2)
        e = e / torch.norm(e, dim=-1, keepdim=True)
        a = m.dec.forward_one_step(e, dec_t, C, dec_states)
        a = a / torch.norm(a, dim=-1, keepdim=True)
        return torch.matmul(e.unsqueeze(-2), a.transpose(-1, -2)).squeeze(-2), dec_t + 1

    t0 = time.time()
    torch.set_num_threads(1)                      # one-frame work: threads only add overhead
    with torch.no_grad():
        for t in range(T):
            e = m.enc.forward_one_step(src[t:t + 1].unsqueeze(0), t, ret_states, caches)
            y, nd = step(e, dec_t)
            if y is not None and dec_t in idx:
                kept[dec_t] = y[0, 0].numpy().copy()
            dec_t = nd
            if t % 3000 == 0:
                print(f"  frame {t}/{T}  {time.time() - t0:.0f} s", flush=True)
        for _ in range(m.delay):
            y, nd = step(torch.zeros(1, 1, m.n_units), dec_t)
            if y is not None and dec_t in idx:
                kept[dec_t] = y[0, 0].numpy().copy()
            dec_t = nd
    dt = time.time() - t0
    rows = np.array(sorted(kept), dtype=np.int64)
    arrays = dict(rows=rows, stream_logits=np.stack([kept[r] for r in rows]))
    p = FX.save_case("ls_hour_stream_c10", ls_meta(m, kind="ls_hour_stream", reference_cpu_seconds=dt, frames_out=dec_t), arrays)
    print(f"ls_hour_stream_c10: reference frame-by-frame over T={T} took {dt:.0f} s ({dt / T * 1e3:.1f} ms/frame); {len(rows)} rows "
          f"-> {os.path.relpath(p)}")


def gen_ls_stream64():
    """The same hour through the ORACLE's frame-by-frame form in float64 (oracle/ls_eend_ref.LsStreamingRef, pinned to the
    reference's streaming driver by tests/golden/ls_stream_T120.npz): the arbiter for how far apart fp32 implementations
    of this recurrence may legitimately be after 36 000 steps -- the reference's own fp32 streaming and batch forms
    differ by ~1e-2 in the logits there (ls_hour_stream_c10 vs ls_hour_c10)."""
    from oracle import ls_eend_ref as R
    m, _ = build_ls()
    T, C = LS_HOUR["T"], LS_HOUR["C"]
    src = FX.make_src([T], 345, LS_HOUR["xseed"])[0].double()
    s = R.LsStreamingRef(m.state_dict(), n_heads=4, enc_n_layers=LS_FULL["enc_n_layers"], dec_n_layers=LS_FULL["dec_n_layers"],
                         conv_kernel_size=LS_FULL["conv_kernel_size"], dtype=torch.float64)
    idx = set(row_index(T).tolist())
    kept, n = {}, 0
    t0 = time.time()
    torch.set_num_threads(1)
    with torch.no_grad():
        for t in range(T):
            y = s.step(src[t].view(1, 1, -1), C)
            if y is not None:
                if n in idx:
                    kept[n] = y[0, 0].numpy().copy()
                n += 1
            if t % 3000 == 0:
                print(f"  frame {t}/{T}  {time.time() - t0:.0f} s", flush=True)
        for _ in range(m.delay):
            y = s.step(None, C)
            if y is not None:
                if n in idx:
                    kept[n] = y[0, 0].numpy().copy()
                n += 1
    rows = np.array(sorted(kept), dtype=np.int64)
    arrays = dict(rows=rows, stream_logits64=np.stack([kept[r] for r in rows]))
    p = FX.save_case("ls_hour_stream64_c10", ls_meta(m, kind="ls_hour_stream64", frames_out=n, cpu_seconds=time.time() - t0), arrays)
    print(f"ls_hour_stream64_c10: oracle float64 frame-by-frame over T={T}: {time.time() - t0:.0f} s -> {os.path.relpath(p)}")


def gen_fs_stream():
    sys.path.insert(0, os.path.join(REF, "FS-EEND"))
    from nnet.model.onl_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import OnlineTransformerDADiarization
    from nnet.model.streaming_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm import StreamingTransformerEDADiarization
    from nnet.utils.copy_params import copy_params_from_masked_to_streaming
    T, C = FS_LONG["T"], FS_LONG["C"]
    torch.manual_seed(FS_LONG["seed"])
    m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, **FS_FULL).eval()
    FX.perturb_(m, FS_LONG["pseed"])
    sm = StreamingTransformerEDADiarization(in_size=345, **FS_FULL).eval()
    copy_params_from_masked_to_streaming(m, sm)
    src = FX.make_src([T], 345, FS_LONG["xseed"])[0]
    idx = row_index(T)
    ys = []
    t0 = time.time()
    with torch.no_grad():
        batch = m.test([src], [T], C)[0][0]
        tb = time.time() - t0
        for t in range(T):
            y = sm.test(src[t].view(1, 1, -1), C)
            if y is not None:
                ys.append(y)
            if t % 500 == 0:
                print(f"  frame {t}/{T}  {time.time() - t0:.0f} s", flush=True)
        for _ in range(m.delay):
            y = sm.test(src[0].view(1, 1, -1), C, dummy_conv_input=True)
            if y is not None:
                ys.append(y)
    ys = torch.cat(ys, dim=1)[0]
    d = (ys - batch).abs().max().item()
    ti = torch.as_tensor(idx)
    meta = dict(kind="fs_stream_long", cfg=FS_FULL, T=T, C=C, seed=FS_LONG["seed"], pseed=FS_LONG["pseed"], xseed=FS_LONG["xseed"],
                in_size=345, checksums=FX.param_checksums(m.state_dict()), torch=torch.__version__, stream_vs_batch_max_abs=d,
                reference_batch_cpu_seconds=tb, reference_stream_cpu_seconds=time.time() - t0 - tb)
    p = FX.save_case("fs_stream_T5000", meta, dict(rows=idx, stream_logits=ys[ti].numpy(), batch_logits=batch[ti].numpy()))
    print(f"fs_stream_T5000: reference streaming vs batch max|d| = {d:.2e} -> {os.path.relpath(p)}")


if __name__ == "__main__":
    {"ls_batch": gen_ls_batch, "ls_stream": gen_ls_stream, "ls_stream64": gen_ls_stream64, "fs_stream": gen_fs_stream}[sys.argv[1]]()
