"""A/B study: weight-gradient kernel (eend_wgrad_bf16 / _bias / conv1d) at the shapes of the FS / LS training steps
(EEND_HIP_LIB selects the library build; tools/ab_variants.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd.train import _call, WS_FLOATS
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
ws = torch.empty(WS_FLOATS, dtype=torch.float32, device=dev)
def timeit(name, fn, flop, byt, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    print(f"{name:34s} {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s  {byt / us / 1e6:6.2f} TB/s algorithmic", flush=True)
for M in (393216, 196608, 32768):
    for N, K in ((256, 256), (768, 256), (1024, 256), (256, 1024), (2048, 256), (256, 2048), (256, 384)):
        dy = (torch.randn(M, N, generator=g) * 1e-3).to(dev).to(torch.bfloat16)
        x = torch.randn(M, K, generator=g).to(dev).to(torch.float16)
        out = torch.empty(N, K, dtype=torch.float32, device=dev)
        bias = torch.empty(N, dtype=torch.float32, device=dev)
        timeit(f"wgrad M={M} N={N} K={K}", lambda: _call("eend_wgrad_bf16", dy, N, x, K, 1, M, N, K, ws, WS_FLOATS, out, K, K, 1.0, 0),
               2.0 * M * N * K, 2.0 * M * (N + K))
        if (N, K) in ((2048, 256), (256, 2048), (256, 256)):
            timeit(f"wgrad+bias M={M} N={N} K={K}", lambda: _call("eend_wgrad_bias_bf16", dy, N, x, K, 1, M, N, K, ws, WS_FLOATS, out, K, K, bias, 1.0, 0),
                   2.0 * M * N * K, 2.0 * M * (N + K))
        if M == 393216 and (N, K) == (256, 256):
            print("   checksum %.6e" % float(out.double().abs().sum()))
        del dy, x
for nseq, Tp in ((64, 512), (64, 1024)):
    M = nseq * Tp
    dy = (torch.randn(M, 256, generator=g) * 1e-3).to(dev).to(torch.bfloat16)
    x = torch.randn(M, 256, generator=g).to(dev).to(torch.float16)
    il = torch.full((nseq,), Tp - 12, dtype=torch.int32, device=dev)
    tmp = torch.empty(256 * 19 * 256, dtype=torch.float32, device=dev)
    gw = torch.empty(256, 256, 19, dtype=torch.float32, device=dev)
    timeit(f"conv wgrad nseq={nseq} Tp={Tp}", lambda: _call("eend_conv1d_wgrad_bf16", dy, x, il, nseq, Tp, 256, 19, 9, ws, WS_FLOATS, tmp, gw),
           2.0 * M * 256 * 4864, 2.0 * M * 512)
