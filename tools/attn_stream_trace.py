"""Perf study: per-item phase timeline of attn_stream.hip (library built with -DEEND_AS_TRACE:
SRC=attn_stream tools/fs_variants.sh astrace=-DEEND_AS_TRACE; run with EEND_HIP_LIB=.../libeend_hip_astrace.so).  Numbers are shader cycles / 100."""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("fs-eend_amd.lib"); ops = importlib.import_module("fs-eend_amd.ops")
dev = "cuda"
g = torch.Generator().manual_seed(0)
Tp, nseq = 512, 384
w = torch.randn(768, 256, generator=g) / 16; b = torch.randn(768, generator=g) * 0.1
w = w.to(dev).half(); b = b.to(dev)
wp = ops.inproj_attn_pack(w)
x = torch.randn(nseq * Tp, 256, generator=g).to(dev).half()
o = torch.empty_like(x)
for _ in range(3):
    ops.inproj_attn_causal_packed(x, wp, b, o, nseq, 4, Tp, 0, 500)
L = _lib.load()
tr = torch.zeros(256 * 6 * 12, dtype=torch.int64, device=dev)
L.eend_debug_attn_stream_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert L.eend_debug_attn_stream_trace(tr.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
t = tr.view(256, 6, 12).cpu().double() / 100.0
NAMES = ["item start", "item0 barrier (X + first weights landed)"] + [f"item{n} barrier" for n in range(1, 6)] + ["projection done", "K/V barrier", "pass 1 (big block) done", "pass 2 done", "end barrier"]
for blk in (0, 255):
    print(f"block {blk}:")
    for ti in range(6):
        print(f"  item {ti}: " + " ".join(f"{v:7.2f}" for v in (t[blk, ti] - t[blk, 0, 0]).tolist()))
dur = t[:, :, 1:] - t[:, :, :-1]
print("mean phase durations [cycles / 100]:")
for k in range(11):
    print(f"  {NAMES[k]:44s} -> {NAMES[k + 1]:44s} {dur[:, :, k].mean():7.2f}  (min {dur[:, :, k].min():.2f}, max {dur[:, :, k].max():.2f})")
print(f"  item period {(t[:, 1:, 0] - t[:, :-1, 0]).mean():.2f}")
