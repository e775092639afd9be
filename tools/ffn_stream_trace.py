"""Perf study: per-tile phase timeline of the packed-stream layer kernel (library built with -DEEND_FS_TRACE:
tools/ab_variants.sh build fstrace=-DEEND_FS_TRACE; run with EEND_HIP_LIB=.../libeend_hip_fstrace.so)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
_lib = importlib.import_module('fs-eend_amd.lib'); ops = importlib.import_module('fs-eend_amd.ops')
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def rn(*s, scale=1.0, dt=torch.float32):
    return (torch.randn(*s, generator=g) * scale).to(dev).to(dt)
wo, bo = rn(256, 256, scale=0.06, dt=torch.float16), rn(256, scale=0.2)
w1, b1 = rn(2048, 256, scale=0.08, dt=torch.float16), rn(2048, scale=0.3)
w2, b2 = rn(256, 2048, scale=0.04, dt=torch.float16), rn(256, scale=0.3)
one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
NAMES = ["tile start", "Wo items done", "LN1 done", "FFN item 0 + activation", "FFN loop done", "last item done", "epilogue done"]
L = _lib.load()
M = 196608
a16, r16 = rn(M, 256, dt=torch.float16), rn(M, 256, dt=torch.float16)
o16 = torch.empty_like(a16)
ws = ops.ffn_stream_pack(wo, w1, w2)
for _ in range(3):
    ops.attnout_ffn_stream(a16, ws, bo, None, r16, one, zero, 1e-5, b1, b2, one, zero, 1e-5, None, o16)
tr = torch.zeros(256 * 8 * 10, dtype=torch.int64, device=dev)
L.eend_debug_fs_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert L.eend_debug_fs_trace(tr.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
raw = tr.view(256, 8, 10).cpu()
t = raw.double() / 100.0                                  # s_memtime ticks are shader cycles here (~2.1 GHz): printed numbers are cycles / 100
nt = M // 192 // 256
for blk in (0, 255):
    b0 = t[blk, 0, 0]
    print(f"block {blk}: tile phases [us since the block's first stamp]")
    for ti in range(nt):
        row = t[blk, ti, :7] - b0
        print(f"  tile {ti}: " + "  ".join(f"{row[k]:7.2f}" for k in range(7)))
d = t[:, :nt, :7]
dur = d[:, :, 1:] - d[:, :, :-1]
print("mean phase durations over all CUs and tiles [us]:")
for k in range(6):
    print(f"  {NAMES[k]:26s} -> {NAMES[k + 1]:26s} {dur[:, :, k].mean():7.2f}  (min {dur[:, :, k].min():.2f}, max {dur[:, :, k].max():.2f})")
print(f"  LN1 done -> item 0 MFMAs done {(t[:, :nt, 7] - t[:, :nt, 2]).mean():.2f}; -> activation done {(t[:, :nt, 3] - t[:, :nt, 7]).mean():.2f}")
print(f"  epilogue done -> next tile start: {(d[:, 1:, 0] - d[:, :-1, 6]).mean():.2f}")
print(f"  tile period: {(d[:, 1:, 0] - d[:, :-1, 0]).mean():.2f}; block total {(d[:, nt - 1, 6] - d[:, 0, 0]).mean():.2f}")
