"""Same-box A/B of the training FFN launches: ffn.hip MODE 3 / 4 (un-packed) vs ffn_train_stream.hip, interleaved, event-timed loops.
usage: python tools/ab_ffn_train.py [M ...]"""
import ctypes, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fs_eend_amd  # noqa
from fs_eend_amd import train as T, lib as L

dev = torch.device("cuda:0")
F16, BF16 = torch.float16, torch.bfloat16


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [196608, 32768, 393216]
    F = 2048
    gen = torch.Generator(device=dev).manual_seed(1)
    for M in Ms:
        for pdrop in (0.1, 0.0):
            x = torch.randn(M, 256, device=dev, generator=gen).to(F16)
            w1 = (torch.randn(F, 256, device=dev, generator=gen) / 16).to(F16)
            b1 = torch.randn(F, device=dev, generator=gen) * 0.1
            w2 = (torch.randn(256, F, device=dev, generator=gen) / math.sqrt(F)).to(F16)
            b2 = torch.randn(256, device=dev, generator=gen) * 0.1
            res = torch.randn(M, 256, device=dev, generator=gen)
            gm, be = torch.ones(256, device=dev), torch.zeros(256, device=dev)
            o32, o16 = torch.empty(M, 256, device=dev), torch.empty(M, 256, dtype=F16, device=dev)
            hid, xh, rs = torch.empty(M, F, dtype=F16, device=dev), torch.empty(M, 256, dtype=F16, device=dev), torch.empty(M, device=dev)
            if pdrop > 0:
                s1 = L.Dropout(12345, int(round(pdrop * (1 << 24))), 1.0 / (1.0 - pdrop))
                s2 = L.Dropout(54321, int(round(pdrop * (1 << 24))), 1.0 / (1.0 - pdrop))
                r1, r2 = ctypes.byref(s1), ctypes.byref(s2)
            else:
                r1 = r2 = None
            n = L.load().eend_ffn_train_stream_elems(F)
            ws = torch.empty(n, dtype=F16, device=dev)
            T._call("eend_ffn_train_stream_pack", w1, w2, ws, F)
            fused = lambda: T._call("eend_ffn_train_f16", x, 256, w1, b1, w2, b2, res, 1.0, gm, be, 1e-5, o32, o16, hid, xh, rs, M, F, r1, r2)
            stream = lambda: T._call("eend_ffn_train_stream_f16", x, 256, ws, b1, b2, res, 1.0, gm, be, 1e-5, o32, o16, hid, xh, rs, M, F, r1, r2)
            r = [(timed(fused), timed(stream)) for _ in range(3)]
            print(f"fwd  M={M} p={pdrop}: fused " + " ".join(f"{a:.1f}" for a, _ in r) + "  stream " + " ".join(f"{b:.1f}" for _, b in r), flush=True)
            if pdrop == 0.0:
                continue
            dy = (torch.randn(M, 256, device=dev, generator=gen) * 1e-4).to(BF16)
            w2t, w1t = w2.t().contiguous().to(BF16), w1.t().contiguous().to(BF16)
            g32 = torch.randn(M, 256, device=dev, generator=gen) * 1e-4
            dh = torch.empty(M, F, dtype=BF16, device=dev)
            wsb = torch.empty(n, dtype=BF16, device=dev)
            T._call("eend_ffn_train_stream_pack", w2t, w1t, wsb, F)
            sc = 1.0 / (1.0 - pdrop)
            fusedb = lambda: T._call("eend_ffn_bwd_data_bf16", dy, 256, w2t, hid, w1t, sc, dh, g32, M, F)
            streamb = lambda: T._call("eend_ffn_bwd_data_stream_bf16", dy, 256, wsb, hid, sc, dh, g32, M, F)
            r = [(timed(fusedb), timed(streamb)) for _ in range(3)]
            print(f"bwd  M={M}: fused " + " ".join(f"{a:.1f}" for a, _ in r) + "  stream " + " ".join(f"{b:.1f}" for _, b in r), flush=True)
            packt = timed(lambda: T._call("eend_ffn_train_stream_pack", w2t, w1t, wsb, F))
            print(f"pack: {packt:.1f} us", flush=True)


main()
