"""Same-box A/B: g += A Wt^T, gemm.hip's tiled kernel (eend_gemm_acc_bf16) against the packed-stream form (gemm_acc_stream.hip)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fs_eend_amd  # noqa
from fs_eend_amd import train as T, lib as L

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[n // 2] * 1e3


for M, K in ((196608, 768), (393216, 1024), (393216, 768), (32768, 768), (65536, 1024), (196608, 256), (196608, 2048)):
    g_ = torch.Generator().manual_seed(1)
    a = (torch.randn(M, K, generator=g_) * 0.5).to(dev).to(BF16)
    wt = (torch.randn(256, K, generator=g_) / 16).to(dev).to(BF16)
    g = torch.randn(M, 256, generator=g_).to(dev)
    ws = torch.empty(L.load().eend_gemm_acc_stream_elems(K), dtype=BF16, device=dev)
    T._call("eend_gemm_acc_stream_pack_bf16", wt, K, ws, K)
    t_old = timeit(lambda: T._call("eend_gemm_acc_bf16", a, K, wt, K, g, 1.0, g, None, M, K))
    g.normal_()
    t_new = timeit(lambda: T._call("eend_gemm_acc_stream_bf16", a, K, ws, g, M, K))
    fl = 2.0 * M * 256 * K
    print(f"[{M}, 256, {K}]: tiled {t_old:.1f} us ({fl / t_old / 1e6:.0f} TFLOP/s) -> stream {t_new:.1f} us ({fl / t_new / 1e6:.0f} TFLOP/s)")
