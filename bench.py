#!/usr/bin/env python
"""bench.py -- FS-EEND frame-wise diarization forward on MI355X.

One "step" = one pass of the hot path (OnlineTransformerDADiarization.test: causal encoder ->
look-ahead conv -> attractor decoder -> head) over one batch of B synthetic 500-frame
utterances already resident in HBM.  Metric = audio frames / s (1 frame = 100 ms of 8 kHz
audio), whole job (all ranks).  Multi-GPU = utterances sharded over ranks, no data-path
collective (weak scaling: per-GPU batch fixed).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

FS_CFG = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, has_mask=True,
              max_seqlen=500, dec_dim_feedforward=2048, mask_delay=0)     # conf/spk_onl_tfm_enc_dec_nonautoreg.yaml
PEAK_MFMA_TFLOPS = 2500.0     # dense bf16/f16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def flops_per_launch(name, shape, T):
    """Algorithmic flops of one launch: row-shaped kernels are priced on the T real
    frames of every sequence, not on the Tp = ceil64(T) slab rows they execute (the 2.4 % padding rows at T = 500 are
    work the layout adds, not work the reference does)."""
    Tp_ = (T + 63) // 64 * 64
    if name not in ("attn_causal", "inproj_attn_causal", "inproj_attn_causal_packed", "retention_chunk", "retention_stream") and shape and shape[0] % Tp_ == 0:
        shape = (shape[0] // Tp_ * T,) + tuple(shape[1:])
    if name == "attn_causal":          # 2*D*T*(T+1) causal-useful flops per sequence, D = H*64
        nseq, H = shape
        return nseq * 2.0 * (H * 64) * T * (T + 1)
    if name in ("inproj_attn_causal", "inproj_attn_causal_packed"):   # packed in-projection (2*Tp*768*256 per sequence) + the causal-useful attention flops
        nseq, H = shape
        Tp = (T + 63) // 64 * 64
        return nseq * (2.0 * (H * 64) * T * (T + 1) + 2.0 * T * 768 * 256)
    if name in ("ffn_fused", "ffn_stream"):
        M, F, K = shape
        return 4.0 * M * F * K
    if name == "fusion_layer_tail":    # 2 out-projections + speaker in-projection + the two FFN GEMMs
        M, F, K = shape
        return 4.0 * M * F * K + 2.0 * M * K * K * 2 + 2.0 * M * 768 * K
    if name == "spk_qkv_attn":         # the in-projection GEMM (the C x C attention itself is VALU work)
        return 2.0 * shape[0] * 768 * 256
    if name == "attnout_spk_stream":   # out-projection + speaker-axis in-projection GEMMs (LayerNorm and attention are VALU work)
        return 2.0 * shape[0] * (256 + 768) * 256
    if name == "attnout_ffn_stream_lo":    # the same with the out-projection's remainder product (the LS-EEND decoder layer tail)
        M, F, K = shape
        return 4.0 * M * F * K + 4.0 * M * K * K
    if name in ("attnout_ffn_fused", "attnout_ffn_stream"):    # out-projection (K x K) + the two FFN GEMMs
        M, F, K = shape
        return 4.0 * M * F * K + 2.0 * M * K * K
    if name == "retention_stream":     # the fused operator: the four projections (2*T*1024*256 per sequence) + the retention core below
        nseq, H, Tv, L = shape
        return nseq * 2.0 * Tv * 1024 * 256 + nseq * H * (Tv // L) * (2 * 64.0 * L * (L + 1) + 2 * 2.0 * L * 64 * 64)
    if name == "retention_chunk":      # (nseq, H, valid frames, L): QK^T + PV causal-useful per chunk, + state build and cross term
        nseq, H, Tv, L = shape
        return nseq * H * (Tv // L) * (2 * 64.0 * L * (L + 1) + 2 * 2.0 * L * 64 * 64)
    if name in ("linear", "linear_res_ln", "linear_res_scale", "linear_res_scale_ln16", "retention_proj", "inproj_heads", "convert_fanout", "conv1d_l2norm", "conv1d_l2norm_stream"):
        M, N, K = shape
        return 2.0 * M * N * K
    return 0.0


def pmc_traffic_file(prefix="pmc_traffic"):
    """(path, first 16 hex digits of its sha256) of the newest committed PMC traffic summary, profiles/rNN_<prefix>.json: the bench line
    names the file it looked the bytes up in (the traffic figure is a repository lookup, not a measurement of this run)."""
    import hashlib
    for r in (6, 5, 4, 3):
        q = os.path.join(ROOT, "profiles", f"r0{r}_{prefix}.json")
        if os.path.exists(q):
            return q, hashlib.sha256(open(q, "rb").read()).hexdigest()[:16]
    return "", None


def pmc_traffic(kernel, shape):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/r04_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH doubled per MI355X_MICROARCH.md).  Only the
    default workload (B=64, T=500, C=6) was profiled; anything else -> None."""
    path = pmc_traffic_file()[0]
    # (alternatives; an alternative that is a tuple needs all its pieces in the kernel name: template lists grew over the rounds)
    tags = {("attnout_ffn_stream", (196608, 2048, 256)): (("ffn_stream_kernel<1, 1, 0, true, 3",), "65536"),
            ("attnout_ffn_stream", (32768, 2048, 256)): (("ffn_stream_kernel<1, 1, 0, true, 2",), "65536"),
            ("attnout_ffn_fused", (196608, 2048, 256)): (("ffn_fused_kernel<1, 0, 1>(FfnParams) #hi", "ffn_fused_kernel<1, 0, true>(FfnParams) #hi"), "131072"),
            ("attnout_ffn_fused", (32768, 2048, 256)): (("ffn_fused_kernel<1, 0, 1>(FfnParams) #lo", "ffn_fused_kernel<1, 0, true>(FfnParams) #lo"), "131072"),
            ("attnout_spk_stream", (196608,)): (("spk_stream_kernel<8",), "65536"),
            ("conv1d_l2norm_stream", (32768, 256, 4864)): (("conv_stream_kernel",), "65536"),
            ("inproj_attn_causal_packed", (64, 4)): ((("inproj_attn_stream_kernel", " #lo"),), "131072"),
            ("inproj_attn_causal_packed", (384, 4)): ((("inproj_attn_stream_kernel", " #hi"),), "131072"),
            ("attn_causal", (64, 4)): (("attn_causal_full_kernel",), "131072"),
            ("attn_causal", (384, 4)): (("attn_causal_full_kernel",), "786432"),
            ("linear_res_ln", (196608, 256, 256)): (("gemm_f16_kernel<64, 256, 1, 4, true, 0, 4", "gemm_f16_kernelIDF16_Li64ELi256ELi1ELi4ELb1ELi0ELi4ELi0E"), "786432")}
    tag = tags.get((kernel, tuple(shape)))
    if tag is None or not os.path.exists(path):
        return None
    for k, v in json.load(open(path))["kernels"].items():
        if any((all(u in k for u in t) if isinstance(t, tuple) else t in k) for t in tag[0]) and k.endswith("grid=" + tag[1]):
            return v["hbm_bytes"]
    return None


def train_pmc_traffic(call, shape, flavour):
    """HBM bytes per launch of a training-step call from the committed PMC passes of one step (profiles/rNN_train_{fs,ls}_pmc_traffic.json,
    tools/gpu_r06_pmc_train.sh; the default workloads only) -> (bytes or None, file, sha256[:16])."""
    path, sha = pmc_traffic_file(f"train_{flavour}_pmc_traffic")
    tags = {"eend_ffn_train_stream_f16": "ffn_train_stream_kernel<1, true, 3>", "eend_ffn_bwd_data_stream_bf16": "ffn_train_stream_kernel<2, false, 3>",
            "eend_attn_causal_bwd_bf16": "attn_bwd_fused_kernel<false, true>", "eend_retention_bwd_bf16": "attn_bwd_fused_kernel<true, false>"}
    big = {"fs": 196608, "ls": 393216}[flavour]
    if call not in tags or not path or not shape or (call.startswith("eend_ffn") and shape[0] != big):
        return None, path, sha
    cand = [v["hbm_bytes"] for k, v in json.load(open(path))["kernels"].items() if tags[call] in k]
    return (max(cand) if cand else None), path, sha


def ls_retention_traffic(nseq, H, Tv, L, fused=False):
    """PMC HBM bytes of the retention kernels of one layer (the default LS workload, 160 x T = 2000, only): the two passes of
    ret_stream.hip + the scan (round 5), or the three kernels of the two-call form (round 3 file)."""
    path = pmc_traffic_file("ls_pmc_traffic")[0] if fused else os.path.join(ROOT, "profiles", "r03_ls_pmc_traffic.json")
    if not os.path.exists(path) or (nseq, H, Tv, L) != (160, 4, 2000, 500):
        return None
    ks = json.load(open(path))["kernels"]
    tot = 0.0
    for name in (("ret_stream_kernel<true>", "ret_state_scan_kernel", "ret_stream_kernel<false>") if fused else
                 ("ret_kv_chunk_kernel", "ret_state_scan_kernel", "ret_chunk_full_kernel")):
        cand = [v["hbm_bytes"] for k, v in ks.items() if name in k]
        if not cand:
            return None
        tot += max(cand)                      # the decoder-layer launch is the larger of the two sizes profiled
    return tot


class OpTimer:
    """Brackets every C-ABI call with HIP events on the launch stream (instrumented steps only)."""

    def __init__(self, ops_mod, T):
        self.ops, self.T, self.rec, self.orig = ops_mod, T, [], {}

    def _wrap(self, name, fn):
        def w(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            if name == "attn_causal":
                shape = (a[4], a[5])
            elif name == "inproj_attn_causal":
                shape = (a[5], a[6])
            elif name == "inproj_attn_causal_packed":   # (x16, w_packed, b_in, o16, nseq, H, Tp, ...)
                shape = (a[4], a[5])
            elif name == "retention_proj":
                shape = (a[0].shape[0], 1024, 256)
            elif name == "retention_chunk":
                tv = k.get("t_valid") or a[11]
                shape = (a[9], a[10], tv // a[12] * a[12], a[12])
            elif name == "retention_stream":            # (x16, xlo16, wstream, bias, o16, st, cscale, sexp, nseq, Tp, chunk, gn_eps, t_valid=)
                tv = k.get("t_valid") or a[9]
                shape = (a[8], 4, tv // a[10] * a[10], a[10])
            elif name in ("linear", "linear_res_ln", "linear_res_scale", "linear_res_scale_ln16"):
                shape = (a[0].shape[0], a[1].shape[0], a[0].shape[1])
            elif name == "ffn_fused":
                shape = (a[0].shape[0], a[1].shape[0], a[0].shape[1])
            elif name == "attnout_ffn_fused":
                shape = (a[0].shape[0], a[7].shape[0], a[0].shape[1])
            elif name == "attnout_ffn_stream":          # (a16, wstream, bo, res, res16, g1, be1, eps1, b1, ...)
                shape = (a[0].shape[0], a[8].shape[0], a[0].shape[1])
            elif name == "attnout_ffn_stream_lo":       # (a16, wstream, bo, res, g1, be1, eps1, b1, ...)
                shape = (a[0].shape[0], a[7].shape[0], a[0].shape[1])
            elif name == "ffn_stream":                  # (x16, wstream, b1, ...)
                shape = (a[0].shape[0], a[2].shape[0], a[0].shape[1])
            elif name == "inproj_heads":
                shape = (a[0].shape[0], a[1].shape[0], a[0].shape[1])
            elif name in ("spk_qkv_attn", "attnout_spk_stream"):
                shape = (a[0].shape[0],)
            elif name == "fusion_layer_tail":
                shape = (a[0].shape[0], a[15].shape[0], 256)
            elif name == "convert_fanout":
                shape = (a[0].shape[0], 256, 256)
            elif name == "conv1d_l2norm":
                shape = (a[0].shape[0], 256, a[1].shape[1])
            elif name == "conv1d_l2norm_stream":        # (x16, wstream, bias, ilens, out32, out16, nseq, Tp, ktaps, pad)
                shape = (a[0].shape[0], 256, a[8] * 256)
            else:
                shape = ()
            self.rec.append((name, shape, s, e))
            return r
        return w

    def __enter__(self):
        for n in ("bn_cast_pad", "gather_bn_cast_pad", "encoder_input", "ffn_fused", "attnout_ffn_fused", "attnout_ffn_stream", "attnout_ffn_stream_lo", "attnout_spk_stream", "ffn_stream", "convert_fanout_f32", "linear", "inproj_heads", "linear_res_ln", "linear_res_scale", "conv1d_l2norm", "conv1d_l2norm_stream",
                  "convert_fanout", "attn_causal", "inproj_attn_causal_packed", "spk_attn", "head_l2dot", "retention_proj", "retention_chunk", "retention_stream", "linear_res_scale_ln16", "linear_glu",
                  "dwconv_bn_swish", "layernorm_f16"):
            if not hasattr(self.ops, n):
                continue
            self.orig[n] = getattr(self.ops, n)
            setattr(self.ops, n, self._wrap(n, self.orig[n]))
        # the f16-residual forms of two entries (same argument positions) are reported under the base kernel's name
        for n, base in (("attnout_ffn_fused_res16", "attnout_ffn_fused"), ("linear_res16_ln", "linear_res_ln")):
            if hasattr(self.ops, n):
                self.orig[n] = getattr(self.ops, n)
                setattr(self.ops, n, self._wrap(base, self.orig[n]))
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.ops, n, f)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, shape, s, e in self.rec:
            key = (name, shape)
            d = agg.setdefault(key, [0.0, 0])
            d[0] += s.elapsed_time(e)
            d[1] += 1
        out = []
        for (name, shape), (ms, n) in agg.items():
            fl = flops_per_launch(name, shape, self.T)
            out.append(dict(kernel=name, shape=list(shape), launches=n, avg_ms=ms / n, total_ms=ms,
                            tflops=(fl / (ms / n * 1e-3) / 1e12) if fl else None))
        return sorted(out, key=lambda d: -d["total_ms"])


def cpu_baseline(T, C, batch=4, budget_s=25.0):
    """The oracle (CPU restatement of the reference, validated against it) timed on the host cores.
    Bounded sample: B=4 utterances per run, a few thread counts, ~25 s of CPU work in total; the best
    thread count is reported (torch oversubscribes badly when given every hardware thread)."""
    from oracle import fs_eend_ref as R
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    torch.manual_seed(0)
    m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, **FS_CFG).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(777)
    src = [torch.randn(T, 345, generator=g) * 2 - 3 for _ in range(batch)]
    ncpu = os.cpu_count() or 1                      # hardware threads; physical cores are reported separately below
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or ncpu
    except Exception:
        phys = ncpu
    cands = sorted({1, min(32, ncpu), phys})          # n = 1 and n = all physical cores (+ 32, torch's sweet spot here)
    t_start, results = time.perf_counter(), {}
    with torch.no_grad():
        for th in cands:
            torch.set_num_threads(th)
            ts = []
            for i in range(4):                      # 1 warm-up + up to 3 timed
                t0 = time.perf_counter()
                R.fs_test(src, [T] * batch, sd, n_heads=4, enc_n_layers=4, dec_n_layers=2, max_nspks=C)
                dt = time.perf_counter() - t0
                if i > 0:
                    ts.append(dt)
                if time.perf_counter() - t_start > budget_s * (cands.index(th) + 1) / len(cands):
                    if not ts:
                        ts.append(dt)
                    break
            ts.sort()
            results[th] = batch * T / ts[len(ts) // 2]
    best = max(results, key=results.get)
    return dict(value=results[best], unit="frames/s", cores=best, kind="port",
                host_physical_cores=phys, host_hw_threads=ncpu, by_threads={str(k): v for k, v in results.items()},
                sample=f"oracle fs_test fp32 on host CPU ({phys} physical cores / {ncpu} hw threads; `cores` = torch threads of the best run), "
                       f"B={batch} x T={T}, C={C}; frames/s by "
                       f"torch threads: " + ", ".join(f"{k}: {v:.0f}" for k, v in results.items()) +
                       f"; {time.perf_counter() - t_start:.1f} s of CPU work")


def batch_sweep(dev, model, T, C, sizes=(1, 512)):
    """the same model.test at B in {1, 64, 512} utterances per GPU (64 is the headline line): frames/s and
    ms/step under hipGraph replay, median of 7 timed replays each.  B = 1 is the latency end (4 + 2*C sequence-heads cannot fill 256
    CUs), B = 512 the throughput end (8x the workgroups of the headline batch)."""
    out = {}
    for Bs in sizes:
        try:
            g = torch.Generator().manual_seed(4242 + Bs)
            src = [(torch.randn(T, 345, generator=g) * 2 - 3).to(dev) for _ in range(Bs)]
            il = [T] * Bs
            model.test(src, il, C)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                model.test(src, il, C)
            torch.cuda.current_stream().wait_stream(st)
            with torch.cuda.graph(gr):
                keep = model.test(src, il, C)
            for _ in range(2):
                gr.replay()
            torch.cuda.synchronize()
            n = 7                                               # median of per-replay event times (see length_sweep)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in evs:
                a.record(); gr.replay(); b.record()
            torch.cuda.synchronize()
            dt = sorted(a.elapsed_time(b) for a, b in evs)[n // 2] * 1e-3
            out[str(Bs)] = dict(frames_per_s=Bs * T / dt, ms_per_step=dt * 1e3, peak_hbm_bytes=int(torch.cuda.max_memory_allocated(dev)))
            del gr, keep, src
            model._ws.clear() if hasattr(model, "_ws") else None
            torch.cuda.empty_cache()
        except Exception as ex:
            out[str(Bs)] = dict(error=f"{type(ex).__name__}: {ex}")
    return out


def length_sweep(dev, model, C, lengths=(300, 500, 1000, 2000), frames_per_step=32000):
    """The same model.test at other chunk lengths (about the same number of frames per step): the packed-weight attention kernel
    (attn_stream.hip) covers the padded length 512 only -- every other length runs the round-2/3 attention kernels (proj.hip + attn_full.hip
    for Tp <= 512, the tiled kernel beyond), so the T = 500 headline does not transfer."""
    out = {}
    for T in lengths:
        Bs = max(1, frames_per_step // T)
        Tp = (T + 63) // 64 * 64
        try:
            g = torch.Generator().manual_seed(977 + T)
            src = [(torch.randn(T, 345, generator=g) * 2 - 3).to(dev) for _ in range(Bs)]
            il = [T] * Bs
            model.test(src, il, C)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                model.test(src, il, C)
            torch.cuda.current_stream().wait_stream(st)
            with torch.cuda.graph(gr):
                keep = model.test(src, il, C)
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            # median of per-replay event times: a one-off stall inside the window (seen in round 5 right after the previous capture's
            # pool was released: 60 ms once, which a mean over 10 replays reported as 8 ms per step for the headline shape) is not the rate
            n = 15
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in evs:
                a.record(); gr.replay(); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            dt = ts[n // 2] * 1e-3
            out[str(T)] = dict(batch=Bs, padded_frames=Tp, frames_per_s=Bs * T / dt, ms_per_step=dt * 1e3, ms_per_step_max=ts[-1],
                               attention_kernel="attn_stream.hip (packed weights)" if Tp <= 512 else "attn_stream.hip (groups of 512 frames + combine)")
            del gr, keep, src
        except Exception as e:                                   # noqa: BLE001
            out[str(T)] = dict(error=str(e)[:200])
        # every length starts from an empty allocator, like every size of batch_sweep
        model._ws.clear() if hasattr(model, "_ws") else None
        torch.cuda.empty_cache()
    return out


def extras(dev):
    """Side measurements (not the headline metric): LS-EEND chunked batch throughput (BASELINE config 3)
    and frame-by-frame streaming latency / real-time factor of both flavours (config 5 mechanism)."""
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    from fs_eend_amd.fs_stream import StreamingTransformerEDADiarization, copy_params_from_masked_to_streaming
    from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization, StreamingConv1d
    res = {}
    LS_CFG = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, max_seqlen=1000,
                  recurrent_chunk_size=500, feed_forward_expansion_factor=4, dec_dim_feedforward=2048,
                  conv_expansion_factor=2, conv_kernel_size=16, half_step_residual=True, conv_delay=9)
    torch.manual_seed(0)
    ls = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **LS_CFG).eval().to(dev)
    g = torch.Generator().manual_seed(1)
    B, T, C = 16, 2000, 10
    src = [(torch.randn(T, 345, generator=g) * 2 - 3).to(dev) for _ in range(B)]
    for _ in range(2):
        ls.test(src, [T] * B, C)
    torch.cuda.synchronize()
    ls_run, ls_launch = (lambda: ls.test(src, [T] * B, C)), "eager launches"
    try:                                         # same hipGraph replay as the headline workload
        gr = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            ls.test(src, [T] * B, C)
        torch.cuda.current_stream().wait_stream(st)
        with torch.cuda.graph(gr):
            ls_static = ls.test(src, [T] * B, C)
        ls_run, ls_launch = gr.replay, "hipGraph replay"
    except Exception:
        torch.cuda.synchronize()
    ls_run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        ls_run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res["ls_eend_batch"] = dict(workload=f"LS-EEND model.test, {B} x T={T} (4 chunks of 500), max_nspks={C}, {ls_launch}",
                                frames_per_s=B * T / dt, ms_per_step=dt * 1e3, rtf=dt / (B * T * 0.1))
    try:                                         # dominant LS kernel, in-situ HIP events over 3 eager steps
        from fs_eend_amd import ops as _ops
        with OpTimer(_ops, T) as tm:
            for _ in range(3):
                ls.test(src, [T] * B, C)
            ks = tm.summary()
        tot = sum(k["total_ms"] for k in ks)
        dom = next(k for k in ks if k["tflops"] is not None)
        res["ls_eend_batch"]["kernel_breakdown"] = [dict(kernel=k["kernel"], shape=k["shape"], launches_per_step=k["launches"] // 3,
                                                         avg_ms=round(k["avg_ms"], 4), share=round(k["total_ms"] / tot, 4),
                                                         tflops=None if k["tflops"] is None else round(k["tflops"], 1)) for k in ks[:8]]
        res["ls_eend_batch"]["instrumented_share_of_step"] = tot / 3 / (dt * 1e3)
        res["ls_eend_batch"]["roofline"] = {"kernel": f"{dom['kernel']} {dom['shape']}", "bound": "mfma", "achieved": dom["tflops"],
                                            "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_MFMA_TFLOPS,
                                            "traffic": None, "avg_launch_ms": dom["avg_ms"]}
        # north_star's "retention chunk recurrence": state build + scan + chunk kernel of one decoder layer, HBM-bound
        # (intensity ~126 flop/B < ridge 312): algorithmic bytes = q, k, v, g read + gated output written, 2-byte elements
        rs = [k for k in ks if k["kernel"] == "retention_stream"]
        if rs:
            # round 5: projections + retention in one operator (ret_stream.hip: chunk K^T V pass, scan, row pass).  SURVEY 8d prices the
            # retention on q, k, v, g read + o written (5 x 512 B per frame and layer) -- tensors this operator never materialises: it
            # reads the input rows (hi + lo f16 halves for the query path) and writes o, 1.5 KB per frame.  `achieved` keeps the SURVEY
            # figure (comparable with rounds 2-4); the fused operator's own algorithmic bytes and the MFMA view are reported beside it.
            r = max(rs, key=lambda k: k["shape"][0])
            nseq_, H_, Tv_, L_ = r["shape"]
            byt = nseq_ * H_ * Tv_ * 64 * 2.0 * 5
            byt_fused = nseq_ * Tv_ * 256 * 2.0 * 3
            # achieved / frac are priced on the bytes the fused operator itself has to move (the SURVEY figure overstated the
            # HBM fraction 1.67x); the SURVEY-comparable rate of rounds 2-4 stays under its own key
            gbs = byt_fused / (r["avg_ms"] * 1e-3) / 1e9
            res["ls_eend_batch"]["roofline_retention"] = {
                "kernel": f"retention_stream (ret_stream<kv> + ret_state_scan + ret_stream<rows>; q/k/v/g projections included) nseq={nseq_} H={H_} T={Tv_} L={L_}",
                "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "avg_launch_ms": r["avg_ms"],
                "mfma_TFLOPs": r["tflops"], "mfma_frac": None if r["tflops"] is None else r["tflops"] / PEAK_MFMA_TFLOPS,
                "algorithmic_bytes_survey": byt, "algorithmic_bytes_fused_operator": byt_fused,
                "survey_comparable_GBs": byt / (r["avg_ms"] * 1e-3) / 1e9,
                "traffic": ls_retention_traffic(nseq_, H_, Tv_, L_, fused=True),
                "traffic_file": os.path.relpath(pmc_traffic_file("ls_pmc_traffic")[0], ROOT) if pmc_traffic_file("ls_pmc_traffic")[0] else None,
                "traffic_file_sha256_16": pmc_traffic_file("ls_pmc_traffic")[1],
                "traffic_source": "repository lookup (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes): the three launches of one "
                                  "decoder layer, bytes per launch",
                "replaces": "retention_proj + retention_chunk (round 4: 0.38 + 0.44 ms for this shape)"}
        rk = [k for k in ks if k["kernel"] == "retention_chunk"]
        if rk and not rs:
            r = max(rk, key=lambda k: k["shape"][0])
            nseq_, H_, Tv_, L_ = r["shape"]
            byt = nseq_ * H_ * Tv_ * 64 * 2.0 * 5
            gbs = byt / (r["avg_ms"] * 1e-3) / 1e9
            res["ls_eend_batch"]["roofline_retention"] = {
                "kernel": f"retention_chunk (ret_kv_chunk + ret_state_scan + ret_chunk_full) nseq={nseq_} H={H_} T={Tv_} L={L_}", "bound": "hbm",
                "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "avg_launch_ms": r["avg_ms"],
                "mfma_TFLOPs": r["tflops"], "mfma_frac": None if r["tflops"] is None else r["tflops"] / PEAK_MFMA_TFLOPS,
                "intensity_flop_per_byte": flops_per_launch("retention_chunk", tuple(r["shape"]), T) / byt,
                "traffic": ls_retention_traffic(nseq_, H_, Tv_, L_),
                "traffic_source": "profiles/r03_ls_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes): ret_kv_chunk + "
                                  "ret_state_scan + ret_chunk_full of one decoder layer, bytes per launch"}
    except Exception as ex:
        res["ls_eend_batch"]["roofline"] = dict(error=f"{type(ex).__name__}: {ex}")

    # BASELINE config 5: one hour of 8 kHz audio (36 000 frames of 100 ms), 8 speakers (+2 slots), processed as
    # 72 chunks of 500 with the retention state carried across chunks -- one model.test call
    try:
        torch.cuda.reset_peak_memory_stats(dev)
        Tl = 36000
        long_src = [(torch.randn(Tl, 345, generator=g) * 2 - 3).to(dev)]
        ls.test(long_src, [Tl], C)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            ls.test(long_src, [Tl], C)
        torch.cuda.synchronize()
        dl = (time.perf_counter() - t0) / 2
        res["ls_eend_longform"] = dict(workload=f"LS-EEND model.test, 1 x T={Tl} (1 h of audio, 72 chunks of 500, state carried), "
                                                f"max_nspks={C}", seconds=dl, rtf=dl / (Tl * 0.1),
                                       peak_hbm_bytes=int(torch.cuda.max_memory_allocated(dev)))
        # the same hour, 8000 frames (16 chunks) at a time with the retention state / conv context carried between
        # calls (ls_model.test_chunked; bit-identical results, tests/test_ls_longform.py)
        ls._ws.clear()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        ls.test_chunked(long_src, [Tl], C)
        torch.cuda.synchronize()
        peak_c = int(torch.cuda.max_memory_allocated(dev) - base)
        t0 = time.perf_counter()
        for _ in range(2):
            ls.test_chunked(long_src, [Tl], C)
        torch.cuda.synchronize()
        dc = (time.perf_counter() - t0) / 2
        res["ls_eend_longform_chunked"] = dict(workload=f"LS-EEND test_chunked, 1 x T={Tl}, super-chunks of 8000 frames, state carried, "
                                                        f"max_nspks={C}; outputs (logits, emb, attractors (T,C,256) f32 = 0.37 GB) included",
                                               seconds=dc, rtf=dc / (Tl * 0.1), peak_activation_bytes=peak_c)
        del long_src
    except Exception as ex:                      # never let a side measurement take the headline number down
        res["ls_eend_longform"] = dict(error=f"{type(ex).__name__}: {ex}")

    # output-side post-processing: threshold + median-11 + segments of a 1-hour activity map,
    # and the DER counters of one evaluation batch, device kernels only (events), next to the reference's CPU path
    try:
        from fs_eend_amd import postproc
        Tl, S = 36000, 8
        xx = torch.randn(Tl + 40, S, generator=g)                # slowly varying tracks: realistic segment structure
        xx = torch.nn.functional.avg_pool1d(xx.t().unsqueeze(0), 41, 1).squeeze(0).t() * 6
        prob = torch.sigmoid(xx + 0.3 * torch.randn(Tl, S, generator=g)).to(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        act = postproc.activity(prob)
        postproc.segments(act)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(20):
            act = postproc.activity(prob)
        ev1.record()
        torch.cuda.synchronize()
        t_act = ev0.elapsed_time(ev1) / 20 * 1e-3
        t0 = time.perf_counter()
        for _ in range(5):
            postproc.make_rttm("rec", prob)
        torch.cuda.synchronize()
        t_rttm = (time.perf_counter() - t0) / 5
        from scipy.signal import medfilt
        pc = prob.cpu()
        t0 = time.perf_counter()
        medfilt(torch.where(pc > 0.5, 1, 0), (11, 1))
        t_cpu = time.perf_counter() - t0
        lg = [torch.randn(500, 6, generator=g).to(dev) for _ in range(64)]
        lb = [(torch.rand(500, 6, generator=g) < 0.3).float().to(dev) for _ in range(64)]
        postproc.der_counters(lg[0], lb[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in zip(lg, lb):
            postproc.der_counters(a, b)
        torch.cuda.synchronize()
        t_der = time.perf_counter() - t0
        res["postprocessing"] = dict(
            workload=f"1-hour activity map T={Tl} x {S} speakers: threshold + median-11 (kernel), full make_rttm (kernels + host "
                     f"formatting); DER counters of 64 x (500 x 6) (64 launches, no host sync)",
            activity_kernel_us=t_act * 1e6, activity_GBps=(Tl * S * 5) / t_act / 1e9, make_rttm_ms=t_rttm * 1e3,
            reference_cpu_threshold_medfilt_ms=t_cpu * 1e3, der_counters_64utt_ms=t_der * 1e3)
    except Exception as ex:
        res["postprocessing"] = dict(error=f"{type(ex).__name__}: {ex}")

    # PIT label assignment: training-batch sized, 64 utterances x 500 frames x 4 speakers
    try:
        from fs_eend_amd import pit
        yl = [torch.randn(500, 4, generator=g).to(dev) for _ in range(64)]
        tl = [(torch.rand(500, 4, generator=g) < 0.4).float().to(dev) for _ in range(64)]
        pit.batch_pit_n_speaker_loss(yl, tl, [4] * 64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pit.batch_pit_n_speaker_loss(yl, tl, [4] * 64)
        torch.cuda.synchronize()
        res["pit_assignment"] = dict(workload="batch_pit_n_speaker_loss, 64 x (500 x 4): cost matrices + assignment + permuted labels, "
                                              "no host round trip for the assignment", ms=(time.perf_counter() - t0) / 10 * 1e3)
    except Exception as ex:
        res["pit_assignment"] = dict(error=f"{type(ex).__name__}: {ex}")

    # feature front-end: one hour of 8 kHz audio -> (36000, 345) log-mel features
    try:
        from fs_eend_amd import feature
        wav = (torch.randn(3600 * 8000, generator=g) * 0.05).to(dev)
        feature.extract_fbank_wave(wav, input_transform="logmel23_cummn")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ft = feature.extract_fbank_wave(wav, input_transform="logmel23_cummn")
        torch.cuda.synchronize()
        tf_ = (time.perf_counter() - t0) / 5
        byt = wav.numel() * 4 + 2 * 360000 * 23 * 4 * 2 + ft.numel() * 4       # wave in, log-mel out/in (x2: normalise), features out
        res["feature_frontend"] = dict(workload="1 h of 8 kHz audio: STFT(200/256/80) -> 23 log-mel -> cumulative-mean norm -> +-7 splice "
                                                "-> /10 subsample, 3 launches", seconds=tf_, rtf=tf_ / 3600.0, out_shape=list(ft.shape),
                                       algorithmic_GBps=byt / tf_ / 1e9)
        del wav, ft
    except Exception as ex:
        res["feature_frontend"] = dict(error=f"{type(ex).__name__}: {ex}")

    # BASELINE config 4: one FS-EEND training step (forward + loss + backward + clip + Adam), 64 x T=500, 4 speakers
    try:
        torch.cuda.reset_peak_memory_stats(dev)
        eng, dtt, loss, _f, _l = time_train(dev, 64, 500, 4, steps=10, warmup=3)
        res["fs_eend_train_step"] = dict(workload="FS-EEND training step, 64 utterances x T=500, 4-speaker mixtures (C=6), shipped yaml "
                                                  f"shapes, dropout {eng.drop_p}, Adam x Noam, clip 5; eager launches, 1 GPU",
                                         ms_per_step=dtt / 10 * 1e3, frames_per_s=64 * 500 * 10 / dtt, final_loss=loss,
                                         peak_hbm_bytes=int(torch.cuda.max_memory_allocated(dev)))
        del eng
        torch.cuda.empty_cache()
    except Exception as ex:
        res["fs_eend_train_step"] = dict(error=f"{type(ex).__name__}: {ex}")

    # ... and its LS half: one LS-EEND training step, 64 x T=1000 (two retention chunks), 4 speakers
    try:
        torch.cuda.reset_peak_memory_stats(dev)
        eng, dtt, loss, _f, _l = time_train(dev, 64, 1000, 4, steps=6, warmup=2, flavour="ls")
        res["ls_eend_train_step"] = dict(workload="LS-EEND training step, 64 utterances x T=1000, 4-speaker mixtures (C=6), shipped yaml "
                                                  f"shapes, dropout {eng.drop_p}, Adam x Noam, clip 5; eager launches, 1 GPU",
                                         ms_per_step=dtt / 6 * 1e3, frames_per_s=64 * 1000 * 6 / dtt, final_loss=loss,
                                         peak_hbm_bytes=int(torch.cuda.max_memory_allocated(dev)))
        del eng, _f, _l
        torch.cuda.empty_cache()
    except Exception as ex:
        res["ls_eend_train_step"] = dict(error=f"{type(ex).__name__}: {ex}")

    # LS-EEND streaming, 8 speakers + 2 slots, O(1) state (LS-EEND/streaming_infer_dia.py:52-97)
    scnn = StreamingConv1d(256, 256, kernel_size=19).to(dev).eval()
    scnn.conv.load_state_dict(ls.cnn.state_dict())
    rs = [dict() for _ in range(4)]
    cc = [torch.zeros(1, 256, 15, device=dev) for _ in range(4)]
    ds = [dict() for _ in range(2)]
    x = src[0]
    nfr, warm = 260, 60
    torch.cuda.synchronize()
    for t in range(nfr):
        if t == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        e = ls.enc.forward_one_step(x[t:t + 1].unsqueeze(0), t, rs, cc)
        e = scnn(e.transpose(1, 2))
        if e is not None:
            e = e.transpose(1, 2)
            e = e / torch.norm(e, dim=-1, keepdim=True)
            a = ls.dec.forward_one_step(e, t, C, ds)
            a = a / torch.norm(a, dim=-1, keepdim=True)
            y = torch.matmul(e.unsqueeze(-2), a.transpose(-1, -2)).squeeze(-2)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / (nfr - warm)
    state_bytes = sum(s["prev_key_value"].numel() * 4 for s in rs + ds) + sum(c.numel() * 4 for c in cc) + 19 * 256 * 4
    res["ls_eend_streaming"] = dict(workload=f"1 stream, max_nspks={C}, frame-by-frame one-step API, eager launches",
                                    ms_per_frame=per * 1e3, rtf=per / 0.1, state_bytes=state_bytes)
    # the same stream through the device-resident session: three hipGraph replays per frame (ls_stream.LsStreamSession)
    try:
        from fs_eend_amd.ls_stream import LsStreamSession
        sess = LsStreamSession(ls, C, batch=1, use_graph=True)
        for t in range(nfr):
            if t == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            sess.push(x[t:t + 1])
        torch.cuda.synchronize()
        per_g = (time.perf_counter() - t0) / (nfr - warm)
        res["ls_eend_streaming_graph"] = dict(workload=f"1 stream, max_nspks={C} (8 speakers + 2 slots), LsStreamSession: state resident in HBM, "
                                                       f"3 hipGraph replays per frame", ms_per_frame=per_g * 1e3, rtf=per_g / 0.1,
                                              speedup_vs_eager=per / per_g)
        nstream = 64                                   # many independent streams per GPU: one session, batch = streams
        sess = LsStreamSession(ls, C, batch=nstream, use_graph=True)
        xb = torch.randn(nfr, nstream, 345, generator=g).to(dev) * 2 - 3
        for t in range(nfr):
            if t == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            sess.push(xb[t])
        torch.cuda.synchronize()
        per_b = (time.perf_counter() - t0) / (nfr - warm)
        res["ls_eend_streaming_graph_64streams"] = dict(workload=f"{nstream} concurrent streams, max_nspks={C}, one session",
                                                        ms_per_frame=per_b * 1e3, rtf_per_stream=per_b / 0.1,
                                                        stream_frames_per_s=nstream / per_b)
    except Exception as ex:
        res["ls_eend_streaming_graph"] = dict(error=f"{type(ex).__name__}: {ex}")

    torch.manual_seed(0)
    fm = OnlineTransformerDADiarization(n_speakers=None, in_size=345, **FS_CFG).eval().to(dev)
    sm = StreamingTransformerEDADiarization(in_size=345, **FS_CFG).eval().to(dev)
    copy_params_from_masked_to_streaming(fm, sm)
    torch.cuda.synchronize()
    for t in range(nfr):
        if t == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        sm.test(x[t].view(1, 1, -1), 6)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / (nfr - warm)
    res["fs_eend_streaming"] = dict(workload=f"1 stream, max_nspks=6, K/V-cache decode attention, frames {warm}..{nfr}",
                                    ms_per_frame=per * 1e3, rtf=per / 0.1)
    try:
        from fs_eend_amd.fs_stream import FsStreamSession
        ses = FsStreamSession(sm, 6, cap=1024, use_graph=True)
        for t in range(nfr):
            if t == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            ses.push(x[t])
        torch.cuda.synchronize()
        per_g = (time.perf_counter() - t0) / (nfr - warm)
        res["fs_eend_streaming_graph"] = dict(workload="1 stream, max_nspks=6, FsStreamSession: K/V caches + device-side history counters "
                                                       "resident in HBM, 3 hipGraph replays per frame (one capture per cache-capacity bucket)",
                                              ms_per_frame=per_g * 1e3, rtf=per_g / 0.1, speedup_vs_eager=per / per_g)
        # the decode step reads the whole K/V history, so its cost grows with the stream position: measured at the
        # positions of BASELINE config 5 (one hour = 36 000 frames; K/V caches 4*1 + 2*6 heads-sets x 2 x t x 256 x 2 B)
        at = {}
        for pos in (5000, 36000):
            ses.seek(pos)
            for t in range(20):
                ses.push(x[t])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(100):
                ses.push(x[20 + t % (nfr - 20)])
            torch.cuda.synchronize()
            at[str(pos)] = (time.perf_counter() - t0) / 100 * 1e3
        res["fs_eend_streaming_graph"]["ms_per_frame_at_t"] = at
        res["fs_eend_streaming_graph"]["rtf_at_t36000"] = at["36000"] * 1e-3 / 0.1
        res["fs_eend_streaming_graph"]["kv_cache_bytes_at_t36000"] = int(sum(c.numel() * 2 for kv in ses.enc_kv + ses.dec_kv for c in kv))
    except Exception as ex:
        res["fs_eend_streaming_graph"] = dict(error=f"{type(ex).__name__}: {ex}")
    return res


def synthetic_labels(lengths, n_spk, seed, dev):
    """Bernoulli(0.3) speaker activity held for 20-frame runs, (T_i, n_spk) float32 on the device."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for T in lengths:
        a = (torch.rand((T + 19) // 20, n_spk, generator=g) < 0.3).float()
        out.append(a.repeat_interleave(20, dim=0)[:T].contiguous().to(dev))
    return out


def train_setup_ls(dev, B, T, n_spk, rank=0):
    """BASELINE config 4, LS half: LS-EEND training step (conf/spk_onl_conformer_retention_enc_dec_nonautoreg.yaml shapes:
    T = 1000 chunks = two retention chunks of 500, 4 Conformer-retention blocks + 2 decoder layers, dropout 0.1), same
    optimiser as FS-EEND (train_dia_simu.py:97-117); SyncBatchNorm statistics are exchanged when world > 1."""
    from fs_eend_amd import config as CFG
    from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization
    from fs_eend_amd.train_ls import LsTrainStep
    from fs_eend_amd.trainer import prepare_labels
    cfg = CFG.load(CFG.LS_EEND_SIMU)
    kw = CFG.model_kwargs(cfg)
    if os.environ.get("EEND_TRAIN_DROPOUT"):
        kw["dropout"] = float(os.environ["EEND_TRAIN_DROPOUT"])
    torch.manual_seed(0)
    model = OnlineConformerRetentionDADiarization(**kw).to(dev).train()
    tr = cfg["training"]
    eng = LsTrainStep(model, warmup=tr["warm_steps"], lr=tr["lr"], schedule_scale=tr["schedule_scale"], grad_clip=tr["grad_clip"],
                      drop_seed=int(tr.get("seed", 0) or 0) * 1000003 + rank)
    g = torch.Generator().manual_seed(777 + rank)
    feats = [(torch.randn(T, 345, generator=g) * 2 - 3).to(dev) for _ in range(B)]
    labels = prepare_labels(synthetic_labels([T] * B, n_spk, 778 + rank, dev), [T] * B)
    return eng, feats, labels


def train_setup(dev, B, T, n_spk, rank=0, flavour="fs"):
    """BASELINE config 4: FS-EEND training step, 4-speaker simulated mixtures, the shipped yaml's shapes and optimiser
    (Adam betas (0.9, 0.98) eps 1e-9 x Noam(warm 100000), clip 5) and its dropout (0.1, all ten sites; counter-hash masks
    applied in the kernels' epilogues, include/eend_hip.h `eend_dropout`).  EEND_TRAIN_DROPOUT overrides the ratio (A/B)."""
    if flavour == "ls":
        return train_setup_ls(dev, B, T, n_spk, rank)
    from fs_eend_amd import config as CFG
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    from fs_eend_amd.train import FsTrainStep
    from fs_eend_amd.trainer import prepare_labels
    cfg = CFG.load(CFG.FS_EEND_SIMU)
    kw = CFG.model_kwargs(cfg)
    if os.environ.get("EEND_TRAIN_DROPOUT"):
        kw["dropout"] = float(os.environ["EEND_TRAIN_DROPOUT"])
    torch.manual_seed(0)
    model = OnlineTransformerDADiarization(**kw).to(dev).train()
    tr = cfg["training"]
    eng = FsTrainStep(model, warmup=tr["warm_steps"], lr=tr["lr"], schedule_scale=tr["schedule_scale"], grad_clip=tr["grad_clip"],
                      drop_seed=int(tr.get("seed", 0) or 0) * 1000003 + rank)
    g = torch.Generator().manual_seed(777 + rank)
    feats = [(torch.randn(T, 345, generator=g) * 2 - 3).to(dev) for _ in range(B)]
    labels = prepare_labels(synthetic_labels([T] * B, n_spk, 778 + rank, dev), [T] * B)
    return eng, feats, labels


def time_train(dev, B, T, n_spk, steps, warmup, rank=0, fence=None, flavour="fs"):
    eng, feats, labels = train_setup(dev, B, T, n_spk, rank, flavour)
    ilens = [T] * B
    for _ in range(warmup):
        eng.step(feats, labels, ilens)
    (fence or torch.cuda.synchronize)()
    gc.collect()
    gc.disable()                      # as timeit does: a generation-2 collection is a 30-40 ms host stall in this process
    t0 = time.perf_counter()
    for _ in range(steps):
        out = eng.step(feats, labels, ilens)
    (fence or torch.cuda.synchronize)()
    dt = time.perf_counter() - t0
    gc.enable()
    return eng, dt, float(out["loss"]), feats, labels


class TrainCallTimer:
    """Brackets every C-ABI call of the training step (fs_eend_amd.train._call and the ops.* forward wrappers it uses)
    with HIP events on the launch stream, for a few instrumented steps after the timed region."""
    INT_ARGS = {"eend_wgrad_bf16": (5, 6, 7), "eend_wgrad_bias_bf16": (5, 6, 7), "eend_gemm_bf16": (7, 8, 9), "eend_gemm_relu_bwd_bf16": (8, 9, 10),
                "eend_linear_relu_train_f16": (7, 8, 9)}
    # ops.* forward wrappers the LS step calls directly (not through _call): name -> shape extractor on the positional args
    OPS = {"linear": lambda a: (a[0].shape[0], a[1].shape[0], a[0].shape[1]),
           "retention_proj": lambda a: (a[0].shape[0], 1024, 256), "convert_fanout": lambda a: (a[0].shape[0], 256, 256),
           # one pass for the forward's f16 operands and the backward's bf16 head rows (proj_stream.hip): (rows, N, 256)
           "proj_stream": lambda a: (int(a[3]), int(a[4]), 256)}

    def __init__(self, T):
        self.T, self.rec = T, []

    def __enter__(self):
        from fs_eend_amd import ops as OPSMOD
        from fs_eend_amd import train as TR
        from fs_eend_amd import train_ls as TL
        self.TR, self.TL, self.OPSMOD, self.orig = TR, TL, OPSMOD, TR._call
        self.orig_ops = {k: getattr(OPSMOD, k) for k in self.OPS}

        def wrap_op(k):
            def w(*a, **kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = self.orig_ops[k](*a, **kw)
                e.record()
                self.rec.append(("ops." + k, tuple(int(x) for x in self.OPS[k](a)), s, e))
                return r
            return w
        for k in self.OPS:
            setattr(OPSMOD, k, wrap_op(k))

        def timed(name, *a):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self.orig(name, *a)
            e.record()
            if name in self.INT_ARGS:
                shape = tuple(int(a[i]) for i in self.INT_ARGS[name])
            elif name == "eend_gemm_acc_bf16":
                shape = (int(a[8]), 256, int(a[9]))
            elif name == "eend_linear_res_ln_train_f16":
                shape = (int(a[14]), 256, int(a[15]))
            elif name == "eend_attn_causal_bwd_bf16":
                shape = (int(a[14]), int(a[15]))
            elif name == "eend_attn_causal_lse_bf16":
                shape = (int(a[5]), int(a[6]))
            elif name == "eend_inproj_heads_train_bf16":
                shape = (int(a[10]) * int(a[11]), 768, 256)
            elif name == "eend_linear_res_scale_ln_train_f16":
                shape = (int(a[14]), 256, int(a[15]))
            elif name == "eend_retention_bwd_bf16":          # (nseq, H, valid frames, chunk)
                shape = (int(a[18]), int(a[19]), int(a[22]), int(a[21]))
            elif name == "eend_retention_chunk_train_f16":
                shape = (int(a[12]), int(a[13]), int(a[19]), int(a[15]))
            elif name == "eend_ffn_train_f16":                # (rows, hidden units, d_model): both linears in one launch
                shape = (int(a[16]), int(a[17]), 256)
            elif name == "eend_ffn_bwd_data_bf16":
                shape = (int(a[8]), int(a[9]), 256)
            elif name == "eend_ffn_train_stream_f16":         # the same two operators on the packed weight stream (round 6)
                shape = (int(a[15]), int(a[16]), 256)
            elif name == "eend_ffn_bwd_data_stream_bf16":
                shape = (int(a[7]), int(a[8]), 256)
            elif name == "eend_gemm_acc_stream_bf16":         # (rows, 256 output features, K) on the packed weight stream (round 6)
                shape = (int(a[4]), 256, int(a[5]))
            else:
                shape = ()
            self.rec.append((name, shape, s, e))
        TR._call = timed
        TL._call = timed
        return self

    def __exit__(self, *exc):
        self.TR._call = self.orig
        self.TL._call = self.orig
        for k, f in self.orig_ops.items():
            setattr(self.OPSMOD, k, f)

    def flops(self, name, shape):
        if name in ("eend_wgrad_bf16", "eend_wgrad_bias_bf16", "eend_gemm_bf16", "eend_gemm_relu_bwd_bf16", "eend_gemm_acc_bf16",
                    "eend_linear_relu_train_f16", "eend_linear_res_ln_train_f16", "eend_inproj_heads_train_bf16",
                    "eend_linear_res_scale_ln_train_f16", "ops.linear", "ops.retention_proj", "ops.convert_fanout"):
            M, N, K = shape
            return 2.0 * M * N * K
        if name in ("eend_ffn_train_f16", "eend_ffn_bwd_data_bf16", "eend_ffn_train_stream_f16", "eend_ffn_bwd_data_stream_bf16"):   # two products of M x F x 256
            M, F_, K = shape
            return 4.0 * M * F_ * K
        if name in ("eend_retention_bwd_bf16", "eend_retention_chunk_train_f16"):
            # per chunk-sequence-head: masked products of L x L x 64 (causal-useful L*(L+1)*64*2 flops each) -- forward 2
            # (QK^T, PV), backward 5 (S, A, dQ, dK, dV) -- plus the cross-chunk terms 2*L*64*64 each (forward 2: state
            # build + cross; backward 5: two outer products, dQ, dK, dV cross terms)   [SURVEY 8d retention row]
            nseq, H_, Tv, L = shape
            nprod = 5.0 if name == "eend_retention_bwd_bf16" else 2.0
            return nseq * H_ * (Tv // L) * nprod * (64.0 * L * (L + 1) + 2.0 * L * 64 * 64)
        if name == "eend_attn_causal_lse_bf16":        # causal-useful flops, 2 products
            return shape[0] * 2.0 * (shape[1] * 64) * self.T * (self.T + 1)
        if name == "eend_attn_causal_bwd_bf16":        # 5 products (S, dP, dQ, dK, dV), causal-useful
            return shape[0] * 5.0 * (shape[1] * 64) * self.T * (self.T + 1)
        return 0.0

    def summary(self, steps):
        torch.cuda.synchronize()
        agg = {}
        for name, shape, s, e in self.rec:
            agg.setdefault((name, shape), []).append(s.elapsed_time(e))
        out = []
        for (name, shape), ts in agg.items():
            n, ms = len(ts), sum(ts)
            fl = self.flops(name, shape)
            row = dict(call=name, shape=list(shape), launches_per_step=n / steps, avg_ms=ms / n, ms_per_step=ms / steps,
                       tflops=(fl / (ms / n * 1e-3) / 1e12) if fl else None)
            med = sorted(ts)[n // 2]
            if max(ts) > 4.0 * med + 0.05:          # an event pair that also covers a host-side stall: say so instead of hiding it
                row["median_ms"], row["max_ms"] = med, max(ts)
                print(f"[bench] {name} {shape}: per-launch ms {['%.3f' % t for t in ts]}", file=sys.stderr)
            out.append(row)
        return sorted(out, key=lambda d: -d["ms_per_step"])


def host_cores():
    ncpu = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or ncpu
    except Exception:
        phys = ncpu
    return phys, ncpu


def cpu_baseline_train(T, n_spk, budget_s=30.0, flavour="fs"):
    """The training oracle (oracle/train_ref.TrainRef / train_ls_ref.LsTrainRef: forward, loss, torch-autograd backward,
    clip, Adam; pinned to the reference's own training_step by tests/golden/{fs,ls}_train_*.npz) timed on the host cores:
    n = all physical cores and n = 1 (the reference's default OMP_NUM_THREADS=1, train_dia.py:4-6), bounded sample."""
    from fs_eend_amd import config as CFG
    cfg = CFG.load(CFG.LS_EEND_SIMU if flavour == "ls" else CFG.FS_EEND_SIMU)
    kw = CFG.model_kwargs(cfg)
    torch.manual_seed(0)
    if flavour == "ls":
        from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization as Model
        from oracle import train_ls_ref as TR
        ocfg = dict(n_units=kw["n_units"], n_heads=kw["n_heads"], enc_n_layers=kw["enc_n_layers"], dec_n_layers=kw["dec_n_layers"],
                    recurrent_chunk_size=kw["recurrent_chunk_size"], conv_delay=kw.get("conv_delay", 9))
        mk = lambda sd: TR.LsTrainRef(sd, ocfg, warmup=cfg["training"]["warm_steps"], clip=cfg["training"]["grad_clip"])
    else:
        from fs_eend_amd.fs_model import OnlineTransformerDADiarization as Model
        from oracle import train_ref as TR
        ocfg = dict(n_units=kw["n_units"], n_heads=kw["n_heads"], enc_n_layers=kw["enc_n_layers"], dec_n_layers=kw["dec_n_layers"],
                    has_mask=kw["has_mask"], mask_delay=kw.get("mask_delay", 0))
        mk = lambda sd: TR.TrainRef(sd, ocfg, warmup=cfg["training"]["warm_steps"], clip=cfg["training"]["grad_clip"])
    m = Model(**kw)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    phys, ncpu = host_cores()
    cands = [phys, 1] if phys > 1 else [1]
    t_start, results, samples = time.perf_counter(), {}, {}
    for ci, th in enumerate(cands):
        batch = 2 if th > 1 else 1
        g = torch.Generator().manual_seed(777)
        feats = [torch.randn(T, 345, generator=g) * 2 - 3 for _ in range(batch)]
        labels = synthetic_labels([T] * batch, n_spk, 778, "cpu")
        torch.set_num_threads(th)
        ref = mk(sd)
        ts = []
        for i in range(4):
            t0 = time.perf_counter()
            ref.step(feats, labels)
            dt = time.perf_counter() - t0
            if i > 0:
                ts.append(dt)
            if time.perf_counter() - t_start > budget_s * (ci + 1) / len(cands):
                if not ts:
                    ts.append(dt)
                break
        ts.sort()
        results[th] = batch * T / ts[len(ts) // 2]
        samples[th] = f"B={batch}, {len(ts)} timed step(s)"
    best = max(results, key=results.get)
    return dict(value=results[best], unit="frames/s", cores=best, kind="port", host_physical_cores=phys, host_hw_threads=ncpu,
                by_threads={str(k): v for k, v in results.items()},
                sample=f"oracle {'LsTrainRef' if flavour == 'ls' else 'TrainRef'}.step fp32 (dropout off) on host CPU, T={T}, {n_spk} "
                       f"speakers; " + "; ".join(f"{k} thread(s): {results[k]:.0f} frames/s ({samples[k]})" for k in results) +
                       f"; {time.perf_counter() - t_start:.1f} s of CPU work")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--slots", type=int, default=6, help="speaker slots C = data.max_speakers + 2")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the LS-EEND / streaming side measurements")
    ap.add_argument("--mode", choices=("infer", "train"), default="infer",
                    help="infer: model.test (BASELINE config 2, the headline metric); train: one optimiser step (config 4)")
    ap.add_argument("--speakers", type=int, default=4, help="train mode: speakers per mixture (labels get +2 columns)")
    ap.add_argument("--flavour", choices=("fs", "ls"), default="fs",
                    help="train mode: fs = FS-EEND (T=500 chunks), ls = LS-EEND (T=1000 chunks unless --frames is given)")
    args = ap.parse_args()
    if args.mode == "train" and args.flavour == "ls" and "--frames" not in " ".join(sys.argv):
        args.frames = 1000                       # data.chunk_size of the LS-EEND yaml

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EEND_DIST_BACKEND", "nccl")       # "gloo": single-GPU rehearsal of the N > 1 path
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from fs_eend_amd import build as _build, ops
    if rank == 0:
        _build.build(verbose=False)
    if world > 1:
        dist.barrier()
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization

    B, T, C = args.batch, args.frames, args.slots

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == "train":
        from fs_eend_amd.shard import job_throughput
        eng, dt, loss, feats, labels = time_train(dev, B, T, args.speakers, args.steps, args.warmup, rank, fence, args.flavour)
        ls = args.flavour == "ls" 
        value = job_throughput(B * T * args.steps, dt, dev)
        dt = world * B * T * args.steps / value
        out = {"metric": f"training audio frames/sec (T={T} chunks)", "value": value, "unit": "frames/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "dtype_detail": "forward: f16 MFMA linears, bf16 QK^T/PV; backward: bf16 MFMA (gradients), f16/bf16 saved activations; "
                               "fp32 accumulate, residual-gradient stream, LayerNorm statistics, Adam on f32 master weights",
               "data": "synthetic",
               "config": {"workload": (f"LS-EEND training step (conf/spk_onl_conformer_retention_enc_dec_nonautoreg.yaml shapes: 4 Conformer-"
                                       f"retention blocks, 2 retention x speaker-attention decoder layers, chunk 500)" if ls else
                                       f"FS-EEND training step (conf/spk_onl_tfm_enc_dec_nonautoreg.yaml shapes)") +
                                      f": {B} utterances/GPU x T={T} x 345, {args.speakers}-speaker mixtures (C={args.speakers + 2} label "
                                      f"columns), forward + BCE/emb-consistency loss + backward + clip 5 + Adam(0.9,0.98,1e-9) x Noam + "
                                      f"operand re-layout, random init (seed 0)",
                          "batch_per_gpu": B, "global_batch": B * world, "frames": T, "dropout": eng.drop_p,
                          "parallelism": f"dp{world}: one all-reduce of the flat {eng.flat.numel * 4 / 1e6:.1f} MB f32 gradient buffer per step",
                          "launch": "eager ctypes launches"},
               "final_loss": loss, "peak_hbm_bytes": int(torch.cuda.max_memory_allocated(dev))}
        if rank == 0 and world == 1 and not args.no_breakdown:
            # dominant kernel: HIP events around every C-ABI call of 2 extra steps (same stream, same inputs).  N = 1 only: a
            # training step contains the gradient all-reduce, a collective every rank would have to enter
            gc.collect()
            gc.disable()              # an event pair must not also cover a host-side collection (seen: 36 ms on one random call)
            with TrainCallTimer(T) as tm:
                for _ in range(2):
                    eng.step(feats, labels, [T] * B)
            calls = tm.summary(2)
            gc.enable()
            tot = sum(c["ms_per_step"] for c in calls)
            out["breakdown"] = [dict(c, share=c["ms_per_step"] / tot) for c in calls[:16]]
            out["kernel_ms_per_step"] = tot
            top = next((c for c in calls if c["tflops"]), None)
            if top is not None:
                tr_bytes, tr_file, tr_sha = train_pmc_traffic(top["call"], top["shape"], args.flavour)
                out["roofline"] = {"kernel": f"{top['call']} {top['shape']}", "bound": "mfma", "achieved": top["tflops"],
                                   "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": top["tflops"] / PEAK_MFMA_TFLOPS,
                                   "traffic": tr_bytes, "traffic_file": os.path.relpath(tr_file, ROOT) if tr_file else None,
                                   "traffic_file_sha256_16": tr_sha, "avg_launch_ms": top["avg_ms"],
                                   "note": "largest time share among the step's MFMA calls; algorithmic flops 2*M*N*K (GEMM family) "
                                           "or 5*D*T*(T+1) per sequence (attention backward); in-situ HIP events on the launch stream"}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_train(T, args.speakers, flavour=args.flavour)
        else:
            out["cpu_baseline"] = None
            out["cpu_baseline_reason"] = ("timed on rank 0 at N = 1 only (the host cores are shared by the ranks of an N > 1 run)" if world > 1
                                          else "--no-cpu-baseline")
        if world > 1:
            # data-parallel sanity of the run just timed: after the same number of optimiser steps every rank must hold the same
            # parameters (one mean all-reduce of the flat gradient per step, identical Adam on every rank)
            cs = torch.stack([eng.flat.params.double().sum(), eng.flat.params.double().abs().sum()]).to(dev)
            allcs = [torch.zeros_like(cs) for _ in range(world)]
            dist.all_gather(allcs, cs)
            out["dp_check"] = {"param_checksums": [[float(v) for v in c.cpu()] for c in allcs],
                               "identical_on_all_ranks": all(bool(torch.equal(c, allcs[0])) for c in allcs)}
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return

    torch.manual_seed(0)
    model = OnlineTransformerDADiarization(n_speakers=None, in_size=345, **FS_CFG).eval().to(dev)
    g = torch.Generator().manual_seed(777 + rank)
    src = [(torch.randn(T, 345, generator=g) * 2 - 3).to(dev) for _ in range(B)]      # resident in HBM
    ilens = [T] * B

    def step():
        return model.test(src, ilens, C)

    step()                                       # allocate the workspace, prepare f16 weights
    torch.cuda.synchronize()
    run = step
    graph_used = False
    if args.graph:
        try:
            gr = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(gr):
                static_out = step()
            run = gr.replay
            graph_used = True
        except Exception as e:                   # capture is an optimisation of launch overhead only
            if rank == 0:
                print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            torch.cuda.synchronize()
            run = step

    for _ in range(args.warmup):
        run()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    dt = time.perf_counter() - t0
    from fs_eend_amd.shard import job_throughput
    value = job_throughput(B * T * args.steps, dt, dev)         # sum of frames / max of time over ranks (RCCL)
    dt = world * B * T * args.steps / value

    out = {
        "metric": "audio frames/sec (T=500 chunks)", "value": value, "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "dtype_detail": "QK^T/PV bf16 MFMA; linear layers + conv f16 MFMA; fp32 accumulate, residual, LayerNorm, softmax",
        "data": "synthetic",
        "config": {"workload": f"FS-EEND conf/spk_onl_tfm_enc_dec_nonautoreg.yaml model.test: {B} utterances/GPU x "
                               f"T={T} frames x 345 log-mel, d_model 256, 4 heads, 4 enc + 2 dec layers, "
                               f"max_nspks={C}, random init (seed 0), eval",
                   "batch_per_gpu": B, "frames": T, "speaker_slots": C, "parallelism": f"utterance-sharded x{world}",
                   "launch": "hipGraph replay" if graph_used else "eager ctypes launches"},
        "rtf": dt / args.steps / (B * T * 0.1),
    }

    if rank == 0 and not args.no_breakdown:
        gc.collect()
        gc.disable()
        with OpTimer(ops, T) as tm:
            for _ in range(3):
                step()
            ksum = tm.summary()
        gc.enable()
        tot = sum(k["total_ms"] for k in ksum)
        out["kernel_breakdown"] = [dict(kernel=k["kernel"], shape=k["shape"], launches_per_step=k["launches"] // 3,
                                        avg_ms=round(k["avg_ms"], 4), share=round(k["total_ms"] / tot, 4),
                                        tflops=None if k["tflops"] is None else round(k["tflops"], 1))
                                   for k in ksum]
        dom = next(k for k in ksum if k["tflops"] is not None)
        fl = flops_per_launch(dom["kernel"], tuple(dom["shape"]), T)
        out["roofline"] = {"kernel": f"{dom['kernel']} {dom['shape']}", "bound": "mfma",
                           "achieved": fl / (dom["avg_ms"] * 1e-3) / 1e12, "peak": PEAK_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": fl / (dom["avg_ms"] * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS,
                           "traffic": pmc_traffic(dom["kernel"], dom["shape"]), "avg_launch_ms": dom["avg_ms"],
                           "traffic_source": "repository lookup: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command (tools/gpu_pmc.sh), bytes per launch",
                           "traffic_file": os.path.relpath(pmc_traffic_file()[0], ROOT) if pmc_traffic_file()[0] else None,
                           "traffic_file_sha256_16": pmc_traffic_file()[1]}
        att = [k for k in ksum if k["kernel"] == "attn_causal" and k["shape"][0] == B]
        # the encoder's time-axis attention launch (nseq = B): the packed-weight form at Tp = 512 (attn_stream.hip), else proj.hip + attn_full.hip / attn.hip
        fus = [k for k in ksum if k["kernel"] in ("inproj_attn_causal_packed", "inproj_attn_causal") and k["shape"][0] == B]
        dec = [k for k in ksum if k["kernel"] in ("inproj_attn_causal_packed", "inproj_attn_causal") and k["shape"][0] == B * C]
        if fus:
            # the encoder's time-axis attention as it runs now: in-projection + QK^T / PV in one kernel, K and V on chip.
            # flops: the packed in-projection (2*Tp*768*256 per sequence, executed on the padded rows) + the causal-useful
            # attention flops 2*D*T*(T+1); `attention_only_*` prices the kernel's WHOLE time against the attention flops
            # alone (the figure comparable with the stand-alone kernel of round 1).
            a = fus[0]
            fl = flops_per_launch(a["kernel"], tuple(a["shape"]), T)
            fl_att = flops_per_launch("attn_causal", tuple(a["shape"]), T)
            Tp_ = (T + 63) // 64 * 64
            byt = a["shape"][0] * (Tp_ * 256 * 2 * 2.0 + 2 * Tp_ * 256 * 2.0)   # X read once + O written once (+ Q scratch round trip, L2)
            tf = fl / (a["avg_ms"] * 1e-3) / 1e12
            out["roofline_attention"] = {
                "kernel": f"encoder {a['kernel']} (in-projection + causal MHA fused) nseq={a['shape'][0]} H=4 T={T}", "bound": "mfma",
                "achieved": tf, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_MFMA_TFLOPS,
                "traffic": pmc_traffic(a["kernel"], a["shape"]), "avg_launch_ms": a["avg_ms"], "mfma_TFLOPs": tf,
                "mfma_frac": tf / PEAK_MFMA_TFLOPS,
                "attention_only_TFLOPs": fl_att / (a["avg_ms"] * 1e-3) / 1e12,
                "attention_only_mfma_frac": fl_att / (a["avg_ms"] * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS,
                "algorithmic_hbm_GBps": byt / (a["avg_ms"] * 1e-3) / 1e9,
                "intensity_flop_per_byte": fl / byt, "ridge_flop_per_byte": PEAK_MFMA_TFLOPS * 1e3 / PEAK_HBM_GBS}
            if dec:        # the decoder's time-axis launch of the same kernel (nseq = B * C): the larger share of the step
                d = dec[0]
                fld, flad = flops_per_launch(d["kernel"], tuple(d["shape"]), T), flops_per_launch("attn_causal", tuple(d["shape"]), T)
                out["roofline_attention"]["decoder_launch"] = {
                    "kernel": f"{d['kernel']} nseq={d['shape'][0]}", "avg_launch_ms": d["avg_ms"],
                    "mfma_frac": fld / (d["avg_ms"] * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS,
                    "attention_only_mfma_frac": flad / (d["avg_ms"] * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS,
                    "traffic": pmc_traffic(d["kernel"], d["shape"])}
        elif att:
            a = att[0]
            fl = flops_per_launch("attn_causal", tuple(a["shape"]), T)
            byt = a["shape"][0] * 4.0 * T * 256 * 2            # Q,K,V read + O written once, 2-byte elements
            # arithmetic intensity of the standalone kernel: 2*D*T*(T+1) flop over 4*T*D*2 B = (T+1)/4 = 125 flop/B at
            # T = 500, below the MI355X ridge (2500 TFLOP/s / 8 TB/s = 312 flop/B): the HBM roof is the binding one,
            # it caps this kernel at 8 TB/s * 125 flop/B = 1.0 PFLOP/s = 40 % of the MFMA peak.
            tf = fl / (a["avg_ms"] * 1e-3) / 1e12
            gbs = byt / (a["avg_ms"] * 1e-3) / 1e9
            out["roofline_attention"] = {
                "kernel": f"encoder attn_causal nseq={a['shape'][0]} H=4 T={T}", "bound": "hbm",
                "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                "traffic": pmc_traffic("attn_causal", a["shape"]), "avg_launch_ms": a["avg_ms"],
                "intensity_flop_per_byte": fl / byt, "ridge_flop_per_byte": PEAK_MFMA_TFLOPS * 1e3 / PEAK_HBM_GBS,
                "mfma_TFLOPs": tf, "mfma_frac": tf / PEAK_MFMA_TFLOPS}

    if rank == 0 and world == 1 and not args.no_extras:
        out["extras"] = extras(dev)
        sw = batch_sweep(dev, model, T, C)
        sw[str(B)] = dict(frames_per_s=value, ms_per_step=dt / args.steps * 1e3)
        out["extras"]["batch_sweep"] = sw
        out["extras"]["length_sweep"] = length_sweep(dev, model, C)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(T, C)
    elif rank == 0:
        out["cpu_baseline"] = None
        out["cpu_baseline_reason"] = ("timed on rank 0 at N = 1 only (the host cores are shared by the ranks of an N > 1 run)" if world > 1
                                      else "--no-cpu-baseline")

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
