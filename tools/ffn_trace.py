"""Perf study: per-tile phase timeline of the fused out-proj + FFN layer kernel (library built with -DEEND_FFN_TRACE:
tools/ab_variants.sh build ffntrace=-DEEND_FFN_TRACE; run with EEND_HIP_LIB=.../libeend_hip_ffntrace.so)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import lib as _lib, ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def rn(*s, scale=1.0, dt=torch.float32):
    return (torch.randn(*s, generator=g) * scale).to(dev).to(dt)
wo, bo = rn(256, 256, scale=0.06, dt=torch.float16), rn(256, scale=0.2)
w1, b1 = rn(2048, 256, scale=0.08, dt=torch.float16), rn(2048, scale=0.3)
w2, b2 = rn(256, 2048, scale=0.04, dt=torch.float16), rn(256, scale=0.3)
one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
NAMES = ["tile start", "A regs -> staging written", "inputs landed (Wo k0/1, A, res)", "A frags read, Wo k2/3 issued", "GEMM half 1 done",
         "Wo k2/3 landed", "GEMM half 2 + LN1 + X staged", "chunk 0 done", "chunk loop done", "epilogue done"]
L = _lib.load()
for M in (196608,):
    x16, res = rn(M, 256, dt=torch.float16), rn(M, 256)
    o32, o16 = torch.empty_like(res), torch.empty_like(x16)
    for _ in range(3):
        ops.attnout_ffn_fused(x16, wo, bo, res, one, zero, 1e-5, w1, b1, w2, b2, one, zero, 1e-5, o32, o16)
    tr = torch.zeros(256 * 16 * 12, dtype=torch.int64, device=dev)
    L.eend_debug_ffn_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert L.eend_debug_ffn_trace(tr.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    t = tr.view(256, 16, 12).cpu().double() / 2400.0          # us at ~2.4 GHz shader clock
    nt = M // 128 // 256
    for blk in (0, 255):
        b0 = t[blk, 0, 0]
        print(f"M={M} block {blk}: tile phases [us since the block's first stamp]")
        for ti in range(nt):
            row = t[blk, ti, :10] - b0
            print(f"  tile {ti}: " + "  ".join(f"{row[k]:7.2f}" for k in range(10)))
    d = t[:, :nt, :10]
    dur = d[:, :, 1:] - d[:, :, :-1]
    print("mean phase durations over all CUs and tiles [us]:")
    for k in range(9):
        print(f"  {NAMES[k]:34s} -> {NAMES[k + 1]:34s} {dur[:, :, k].mean():7.2f}  (min {dur[:, :, k].min():.2f}, max {dur[:, :, k].max():.2f})")
    gap = d[:, 1:, 0] - d[:, :-1, 9]
    print(f"  epilogue done -> next tile start: {gap.mean():.2f}")
    print(f"  tile period: {(d[:, 1:, 0] - d[:, :-1, 0]).mean():.2f}; block total {(d[:, nt - 1, 9] - d[:, 0, 0]).mean():.2f}")
